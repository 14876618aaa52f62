// logger.h -- minimal stderr logger standing in for the reference's abseil wrapper
// (src/logger.h:4-15, src/logger.cpp:5-12).  LOG(INFO) is off by default: the reference logs one
// line per layer per Forward (src/layer.cpp:46), which is in the hot loop.  SI_LOG_LEVEL=0 enables it.
#ifndef SIMPLE_INFER_SRC_LOGGER_H_
#define SIMPLE_INFER_SRC_LOGGER_H_

#include <iostream>
#include <sstream>

namespace SimpleInfer {

enum LogSeverity { INFO = 0, WARNING = 1, ERROR = 2 };

int LogThreshold();
void InitializeLogger();

// only ever constructed for an enabled severity (see LOG below)
class LogLine {
public:
    LogLine(LogSeverity sev, const char* file, int line) {
        os_ << "[simpleinfer " << (sev == INFO ? "I" : sev == WARNING ? "W" : "E") << " " << file << ":" << line << "] ";
    }
    ~LogLine() { std::cerr << os_.str() << std::endl; }
    template<typename T>
    LogLine& operator<<(const T& v) {
        os_ << v;
        return *this;
    }

private:
    std::ostringstream os_;
};

// swallows the stream expression of a disabled LOG so that both arms of the conditional have type void
struct LogVoidify {
    void operator&(const LogLine&) const {}
};

}  // namespace SimpleInfer

// Short circuit: with the severity below the threshold neither the stream nor any operand of << is evaluated (the
// reference logs one line per layer per Forward, src/layer.cpp:46 -- that sits on the launch path).
#define LOG(sev)                                                       \
    (::SimpleInfer::sev < ::SimpleInfer::LogThreshold()) ? (void)0     \
        : ::SimpleInfer::LogVoidify() & ::SimpleInfer::LogLine(::SimpleInfer::sev, __FILE__, __LINE__)

#endif
