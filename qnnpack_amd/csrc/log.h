/*
 * log.h -- leveled stderr logging for the host code. Role-equivalent to the
 * reference's src/qnnpack/log.h:15-23 (clog wrappers): error paths of
 * create/setup log one line and return an enum qnnp_status; the run path never
 * logs. Level from env QNNP_LOG_LEVEL (0 none, 1 error [default], 2 warning,
 * 3 info, 4 debug).
 */
#pragma once

#include <inttypes.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

static inline int qnnp_log_level(void) {
  static int level = -1;
  if (level < 0) {
    const char* env = getenv("QNNP_LOG_LEVEL");
    level = env != NULL ? atoi(env) : 1;
  }
  return level;
}

static inline void qnnp_vlog(int level, const char* tag, const char* fmt, va_list ap) {
  if (qnnp_log_level() < level) return;
  fprintf(stderr, "%s in QNNPACK-gfx950: ", tag);
  vfprintf(stderr, fmt, ap);
  fputc('\n', stderr);
}

#define QNNP_DEFINE_LOG(name, level, tag)                      \
  static inline void name(const char* fmt, ...) {              \
    va_list ap;                                                \
    va_start(ap, fmt);                                         \
    qnnp_vlog(level, tag, fmt, ap);                            \
    va_end(ap);                                                \
  }

QNNP_DEFINE_LOG(qnnp_log_error, 1, "Error")
QNNP_DEFINE_LOG(qnnp_log_warning, 2, "Warning")
QNNP_DEFINE_LOG(qnnp_log_info, 3, "Note")
QNNP_DEFINE_LOG(qnnp_log_debug, 4, "Debug")
