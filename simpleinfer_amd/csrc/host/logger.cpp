#include "logger.h"

#include <cstdlib>

namespace SimpleInfer {

static int g_threshold = -1;

int LogThreshold() {
    if (g_threshold < 0) {
        const char* e = std::getenv("SI_LOG_LEVEL");
        g_threshold = e ? std::atoi(e) : (int)WARNING;
    }
    return g_threshold;
}

void InitializeLogger() { (void)LogThreshold(); }

}  // namespace SimpleInfer
