// STUB (see Reference.h): PSRCHIVE's exception class, signatures only
#ifndef STUB_Error_h
#define STUB_Error_h
#include <stdarg.h>
#include <string>
#include <iostream>
enum ErrorCode { Undefined, BadAllocation, BadPointer, InvalidParam, InvalidState, InvalidRange, FileNotFound, FailedCall,
                 FailedSys, EndOfFile };
class Error {
public:
  Error (ErrorCode c, std::string func, const char* msg = 0, ...);
  Error (ErrorCode c, std::string func, std::string msg);
  virtual ~Error ();
  const Error& operator+= (const char* func);
  const Error& operator+= (const std::string& func);
  const std::string get_message () const;
  ErrorCode get_code () const;
  virtual void report (std::ostream&) const;
};
std::ostream& operator<< (std::ostream&, const Error&);
#endif
