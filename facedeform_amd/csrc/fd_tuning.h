// fd_tuning.h -- the one place environment variables are read.
#pragma once
// Tuning switches.  The A/B switches of rounds 1-4 (FD_SOLVER, FD_REG_BUILD, FD_NO_GRAPH, FD_SHARED_*, FD_NO_BALANCE ...) are
// environment variables ONLY in tuning builds (-DFD_TUNING: tools/, profile collections); the product library compiles every
// one of them to its default -- a library that lives inside Houdini does not change its solver or its launch shape by the
// environment of the process that loaded it.  Per-context choices go through fd_config.
#include <cstdlib>
#ifdef FD_TUNING
static inline const char *tuning_env(const char *name) { return getenv(name); }
#else
static inline const char *tuning_env(const char *) { return nullptr; }
#endif

