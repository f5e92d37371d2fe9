// oracle/prelude_threads.h — TEST INFRASTRUCTURE: replacement for the reference's force-included prelude (src/common.h)
// for the CPU-baseline builds.  src/common/config.h hard-codes NUM_HYPERTHREADS = 16 (:23), sizes the worker arrays of
// Render_Waiter with it (src/rt/renderer.c:205-209) and has no #ifndef guards, so -DNUM_HYPERTHREADS=... cannot override
// it.  This prelude does what common.h does — platform macro, the configuration values, common/macros.h,
// common/types.h — with config.h's values restated and the thread count taken from -DRTX_THREADS=N.
// No reference source includes common.h or common/config.h by name.
#pragma once

#if defined(__linux__)
#    define TARGET_LINUX
#endif

#ifndef RTX_THREADS
#    error "compile with -DRTX_THREADS=<worker threads>"
#endif
#define RTX_ULL_(x) x##ull
#define RTX_ULL(x)  RTX_ULL_(x)

#define RENDER_FPS        (5)
#define RENDER_TILE_W_PX  (2)
#define RENDER_TILE_H_PX  (2)
#define RT_EPSILON        (0.0001f)
#define SZ_CACHE_LINE     (64ull)
#define NUM_HYPERTHREADS  (RTX_ULL(RTX_THREADS))
#define SZ_L1_CACHE_TOTAL (512ull * 1024ull)
#define SZ_L1_CACHE       (SZ_L1_CACHE_TOTAL / NUM_HYPERTHREADS)

#include "common/macros.h"
#include "common/types.h"
