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
  refcountable_ptr(T* p = NULL) : the_p(p) { inc(); }
  refcountable_ptr(const refcountable_ptr& rhs) : the_p(rhs.the_p) { inc(); }
  virtual ~refcountable_ptr() { dec(); }
  refcountable_ptr& operator=(const refcountable_ptr& rhs) {
    if (the_p != rhs.the_p) { dec(); the_p = rhs.the_p; inc(); }
    return *this;
  }
  T& operator*() const { return *static_cast<T*>(the_p); }
  T* operator->() const { return static_cast<T*>(the_p); }
  bool is_null() const { return the_p == NULL; }
  bool unique() const { return the_p && the_p->unique(); }
  friend bool operator==(const refcountable_ptr& a, const refcountable_ptr& b) { return a.the_p == b.the_p; }
  friend bool operator!=(const refcountable_ptr& a, const refcountable_ptr& b) { return a.the_p != b.the_p; }
 protected:
  Countable* the_p;
 private:
  void inc() { if (the_p) the_p->increment(); }
  void dec() { if (!the_p) return; if (the_p->unique()) delete the_p; else the_p->decrement(); the_p = NULL; }
};

// smart pointer with the inheritance of the object pointed to
template <class DerivedType, class BaseTypePtr>
class Inherit : public BaseTypePtr {
 public:
  Inherit(DerivedType* s = NULL) : BaseTypePtr(s) {}
  DerivedType* operator->() const { return static_cast<DerivedType*>(BaseTypePtr::the_p); }
};
