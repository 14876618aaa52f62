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

class LogLine {
public:
    LogLine(LogSeverity sev, const char* file, int line) : on_(sev >= LogThreshold()) {
        if (on_) os_ << "[simpleinfer " << (sev == INFO ? "I" : sev == WARNING ? "W" : "E") << " " << file << ":" << line << "] ";
    }
    ~LogLine() {
        if (on_) std::cerr << os_.str() << std::endl;
    }
    template<typename T>
    LogLine& operator<<(const T& v) {
        if (on_) os_ << v;
        return *this;
    }

private:
    bool on_;
    std::ostringstream os_;
};

}  // namespace SimpleInfer

#define LOG(sev) ::SimpleInfer::LogLine(::SimpleInfer::sev, __FILE__, __LINE__)

#endif
