// Stable LSD radix sort of (u64 key, u32 value) pairs, 8 bits per pass, hand-written for wave64.
// Used by unique-rows (K3), segment plans and the strided-conv output-coordinate unique (K8).
// One launch per pass (one-sweep: global digit histograms upfront + decoupled look-back over per-tile digit counts); only
// the passes covering [0, key_bits) are run.
#pragma once
#include "common.h"

namespace fsf {

constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;  // 2048 keys per workgroup
constexpr int RS_BINS = 256;

static inline int64_t radix_grid_tiles(int64_t n) { return n > 0 ? (n + RS_TILE - 1) / RS_TILE : 1; }
// tiles' worth of 256-word scratch the sorter's `hist` buffer is sized by: the four-launch form needs tiles + 1, the one-sweep
// form (8 passes at most) passes * tiles status blocks + the global histograms + the tickets
static inline int64_t radix_num_tiles(int64_t n) { return 8 * radix_grid_tiles(n) + 10; }
// bytes of scratch the sorter needs (alternate key/value buffers + per-tile digit histograms)
int64_t radix_sort_scratch_bytes(int64_t n);

// Sorts n pairs by the low `key_bits` bits of the key.  keys_a/vals_a hold the input and are clobbered;
// keys_b/vals_b are the alternate buffers.  On return *keys_out/*vals_out point at whichever buffer holds
// the sorted result.  `hist` = u32[RS_BINS * (radix_num_tiles(n) + 1)]; `hist_zeroed`: the caller has cleared ALL of it on `stream`
// already (one memset over several scratch arrays instead of one per helper).
int radix_sort_pairs(uint64_t* keys_a, uint32_t* vals_a, uint64_t* keys_b, uint32_t* vals_b, uint32_t* hist,
                     int64_t n, int key_bits, uint64_t** keys_out, uint32_t** vals_out, hipStream_t stream, bool hist_zeroed = false);

}  // namespace fsf
