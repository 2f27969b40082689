// common/refcount.h -- intrusive reference counting with the reference's names
// (reference common/refcount.h:21-32, 171-270): Countable, refcountable_ptr<T>, Inherit<Derived, BasePtr>.
#pragma once
#include <cassert>
#include <climits>
#include <cstddef>
#include <string>
#include "common/jexception.h"

typedef std::string String;

class Countable {
 public:
  virtual ~Countable() {}
  bool unique() const { return count_ == 1; }
  void increment() { count_++; }
  void decrement() { assert(count_ > 0); count_--; }
 protected:
  Countable() : count_(0) {}
 private:
  unsigned count_;
};

template <class T>
class refcountable_ptr {
 public:
  refcountable_ptr(T* p = NULL) : the_p(p), smart_behavior_(true) { increment(); }
  refcountable_ptr(const refcountable_ptr& rhs) : the_p(rhs.the_p), smart_behavior_(true) { increment(); }
  virtual ~refcountable_ptr() { if (smart_behavior_) decrement(); }
  // give up ownership: the pointer keeps pointing at the object but no longer counts (reference refcount.h:204-214)
  void disable() {
    if (is_null()) throw jconsistency_error("Attempted to disable a NULL pointer.");
    if (unique()) throw jconsistency_error("Attempted to disable a unique pointer.");
    smart_behavior_ = false;
    decrement();
  }
  refcountable_ptr& operator=(const refcountable_ptr& rhs) {
    if (the_p != rhs.the_p) {
      if (smart_behavior_) decrement();
      the_p = rhs.the_p;
      increment();
    }
    return *this;
  }
  refcountable_ptr& operator=(T* rhs) {            // (reference refcount.h:227-236; re-enables a disabled pointer)
    if (static_cast<Countable*>(rhs) == the_p && smart_behavior_) return *this;
    if (smart_behavior_) decrement(); else smart_behavior_ = true;
    the_p = rhs;
    increment();
    return *this;
  }
  T& operator*() const { return *static_cast<T*>(the_p); }
  T* operator->() const { return static_cast<T*>(the_p); }
  bool is_null() const { return the_p == NULL; }
  bool unique() const { return the_p && the_p->unique(); }
  void increment() { if (the_p) the_p->increment(); }
  void decrement() { if (!the_p) return; if (the_p->unique()) delete the_p; else the_p->decrement(); }
  friend bool operator==(const refcountable_ptr& a, const refcountable_ptr& b) { return a.the_p == b.the_p; }
  friend bool operator!=(const refcountable_ptr& a, const refcountable_ptr& b) { return a.the_p != b.the_p; }
 protected:
  Countable* the_p;
 private:
  bool smart_behavior_;
};

// smart pointer with the inheritance of the object pointed to
template <class DerivedType, class BaseTypePtr>
class Inherit : public BaseTypePtr {
 public:
  Inherit(DerivedType* s = NULL) : BaseTypePtr(s) {}
  DerivedType* operator->() const { return static_cast<DerivedType*>(BaseTypePtr::the_p); }
};
