/* Harness shim so that the reference's sample mains build unchanged against the compat
 * headers: monotonic timer(), __rdtsc, ALIGNSPEC.  Not part of the coder. */
#ifndef RYG_RANS_AMD_COMPAT_PLATFORM_H
#define RYG_RANS_AMD_COMPAT_PLATFORM_H
#ifndef __STDC_FORMAT_MACROS
#define __STDC_FORMAT_MACROS
#endif
#include <assert.h>
#include <inttypes.h>
#include <time.h>
#if defined(__x86_64__) || defined(__i386__)
#include <x86intrin.h>
#endif
#define ALIGNSPEC(type, name, alignment) type name __attribute__((aligned(alignment)))
static inline double timer()
{
    struct timespec now;
    clock_gettime(CLOCK_MONOTONIC, &now);
    return (double)now.tv_sec + (double)now.tv_nsec * 1e-9;
}
#endif
