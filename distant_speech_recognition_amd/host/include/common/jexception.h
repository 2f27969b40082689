// common/jexception.h -- the reference's exception family (reference common/jexception.h:26-161):
// printf-style messages, an error_type code, jiterator_error at end of stream.
#pragma once
#include <cstdarg>
#include <cstdio>
#include <exception>
#include <string>

typedef enum { JERROR, JALLOCATION, JARITHMETIC, JCONSISTENCY, JDIMENSION, JINDEX, JINITIALIZATION, JIO,
               JITERATOR, JPYTHON, JKEY, JNUMERIC, JPARAMETER, JPARSE, JTYPE } error_type;

class j_error : public std::exception {
 public:
  j_error() throw() : _what(""), code(JERROR) {}
  j_error(const char* what_arg, ...) throw() : code(JERROR) { va_list ap; va_start(ap, what_arg); format(what_arg, ap); va_end(ap); }
  virtual ~j_error() throw() {}
  virtual const char* what() const throw() { return _what.c_str(); }
  error_type getCode() { return code; }
 protected:
  void format(const char* fmt, va_list ap) { char buf[1024]; vsnprintf(buf, sizeof(buf), fmt, ap); _what = buf; }
  std::string _what;
  error_type code;
};

#define BTK_DEFINE_JERROR(NAME, CODE)                                                        \
  class NAME : public j_error {                                                              \
   public:                                                                                   \
    NAME(const char* what_arg, ...) { va_list ap; va_start(ap, what_arg); format(what_arg, ap); va_end(ap); code = CODE; } \
  };
BTK_DEFINE_JERROR(jallocation_error, JALLOCATION)
BTK_DEFINE_JERROR(jarithmetic_error, JARITHMETIC)
BTK_DEFINE_JERROR(jconsistency_error, JCONSISTENCY)
BTK_DEFINE_JERROR(jdimension_error, JDIMENSION)
BTK_DEFINE_JERROR(jindex_error, JINDEX)
BTK_DEFINE_JERROR(jinitialization_error, JINITIALIZATION)
BTK_DEFINE_JERROR(jio_error, JIO)
BTK_DEFINE_JERROR(jiterator_error, JITERATOR)
BTK_DEFINE_JERROR(jkey_error, JKEY)
BTK_DEFINE_JERROR(jnumeric_error, JNUMERIC)
BTK_DEFINE_JERROR(jparameter_error, JPARAMETER)
BTK_DEFINE_JERROR(jparse_error, JPARSE)
BTK_DEFINE_JERROR(jtype_error, JTYPE)
