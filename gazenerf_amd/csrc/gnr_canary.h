// gnr_canary.h -- carve-internal canaries (round 6).  EXPERIMENTAL BUILDS ONLY: -DGNR_CANARY=1.
//
// The guard-band allocator of tests/guard/ sees a write that leaves a buffer the CALLER owns.  The out-of-bounds write round 5 found
// in gnr_wgrad.hip (commit 1503d29) did not: the rider shares of one weight-gradient GEMM ran into the partial tiles of the next,
// INSIDE the caller's scratch -- every workspace of this library is one caller-owned buffer that the entry point carves into
// regions (carve_fwd, carve_bwd, up_carve, up_carve_bwd, the weight-gradient arena and the three sub-blocks of a GEMM's scratch).
// With GNR_CANARY the carve functions leave CANARY_BYTES between consecutive regions, the entry point fills those gaps with a pattern
// before its first launch and compares them after its last (that build synchronises the stream and reads one counter back: timing
// and the no-host-sync rule do not apply to it), and a hit fails the call naming the carve and the region index.  gnr_bwd checks
// the gaps of the SAVED forward workspace without refilling them: neither pass may have touched them.
// Product builds: CANARY_BYTES == 0 and every hook below is an empty inline -- the carved layouts are byte for byte the old ones.
// How to run it: tools/session.sh <name> canary (rebuilds with the flag, runs the fuzzers and the parity tests, restores).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

#if defined(GNR_CANARY) && !defined(GNR_EXPERIMENTAL_BUILD)
#error "GNR_CANARY needs -DGNR_EXPERIMENTAL_BUILD (python -m gazenerf_amd.build adds it when GNR_EXTRA_HIPCC_FLAGS is set)"
#endif

namespace gnr {

#ifdef GNR_CANARY
constexpr size_t CANARY_BYTES = 4096;
// Start a call: forget this thread's gaps.  fill: gaps noted from now on are filled by canary_arm (false: only checked).
void canary_begin(bool fill);
void canary_fill_mode(bool fill);
// A gap of CANARY_BYTES at `gap` behind region number `index` of carve `what` (called by the carve functions when they are given a base).
void canary_note(void* gap, const char* what, int index);
// The same for a gap that comes into being during the call (sub-allocations of the weight-gradient arena): filled right away on `st`.
void canary_note_now(void* gap, const char* what, int index, hipStream_t st);
void canary_arm(hipStream_t st);                          // fills the gaps noted in fill mode
int canary_check(hipStream_t st, const char* entry);      // synchronises; 0 = every gap intact, else fail(...) and non-zero
#else
constexpr size_t CANARY_BYTES = 0;
inline void canary_begin(bool) {}
inline void canary_fill_mode(bool) {}
inline void canary_note(void*, const char*, int) {}
inline void canary_note_now(void*, const char*, int, hipStream_t) {}
inline void canary_arm(hipStream_t) {}
inline int canary_check(hipStream_t, const char*) { return 0; }
#endif

}  // namespace gnr
