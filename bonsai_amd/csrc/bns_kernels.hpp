// bns_kernels.hpp -- kernel parameter block shared by the kernels and the host-side launcher.
#pragma once
#include "bns_device.hpp"

namespace bns {

// Units a wavefront claims at a time.  One 64-lane load brings a chunk's offsets (chunk * mates + 1 of them: at most 63 reads or
// 31 pairs), and all claims go to ONE counter whose atomics saturate at about 80 M/s: at 1.2 G reads/s chunks of 8 or 16 units
// are bound by that (15.0 / 8.3 ms per 10 M reads), chunks of 31 are not (7.0 ms); 47 and 63 balance a little worse (7.05 / 7.12).
#ifndef BNS_CLASSIFY_CHUNK1
#define BNS_CLASSIFY_CHUNK1 31
#endif
#ifndef BNS_CLASSIFY_CHUNK2
#define BNS_CLASSIFY_CHUNK2 31
#endif
constexpr u32 classify_chunk(u32 nmates) { return nmates == 1 ? BNS_CLASSIFY_CHUNK1 : BNS_CLASSIFY_CHUNK2; }
constexpr u32 LDS_CAP = 128;   // distinct taxa per unit held in LDS; beyond that the overflow kernel takes over

struct ClassifyParams {
    // reads: ASCII (classify packs in-kernel) and pre-packed words (encode / build kernels)
    const u8 *bases;
    const u64 *words;
    const u32 *nmask;
    const u64 *offsets;
    u64 n_units;
    int nmates;
    // table: bucket layout
    const Slot *slots;          // BUCKET layout table, or the overflow table of the MINBUCKET layout
    const MinBucket *minb;
    u64 bucket_mask;            // BUCKET layout: bucket count - 1 (a power of two)
    u32 n_mb;                   // MINBUCKET layout: buckets a key can call home (any count; MINB_MAX_CHAIN - 1 more follow for spills)
    u64 ovf_mask;               // bucket mask of the MINBUCKET overflow table (p.slots)
    // table: khash layout (on-disk arrays)
    const u32 *kflags;
    const u64 *kkeys;
    const u32 *kvals;
    u64 kh_nb;
    // taxonomy
    const TaxNode *nodes;
    u32 n_nodes;
    // encoder
    u32 k, c;
    u32 m;              // minimizer length of the MINBUCKET layout (== k: plain hashing)
    u32 min_wide;                        // the table was built with the wide minimizer identity (MinSpec::wide; selects the kernel)
    u32 min_len, min_shift, min_canon;   // where in the key the minimizer is taken (MinSpec): whole key, canonical, for contiguous
                                         // seeds; the mask's longest run of adjacent sampled bases, plain m-mers, for spaced ones
    u32 w;              // Spacer window (bases); w <= c means unwindowed.  Only encode / build honour it (classify is w = k)
    int score;          // BNS_SCORE_* for windowed minimizer selection
    double ent_tbl[33]; // BNS_SCORE_ENTROPY_STRING: (n/k) ln(n/k) for n = 0..k, computed by the host's libm
    u64 *win_scratch;   // emitted-stream windows wider than 64 k-mers: per-wavefront queue image, 2 * (ws + 64) u64 each
    int canon;
    int dbg;            // ablation bits for profiling only (bns_debug_set); 0 in production
    int want_hits;      // hits != nullptr (hot copy; the pointer itself is a cold argument)
    int emit_none;      // reference behaviour for a spaced seed through the string for_each: no k-mers
    u16 pos[32];        // cumulative offsets of the k sampled bases (pos[0] = 0)
    // spaced seeds with comb <= 64: the mask as runs of adjacent sampled bases (fast gather from an aligned window)
    u32 n_runs;
    u8 run_start[32], run_len[32];
    u64 sample_mask;    // bit (63-i) set when base i of the comb is sampled
    // masks with many runs (e.g. 1x15,0x15: fifteen one-base runs and one of sixteen): the sampled 2-bit fields of the two
    // 32-base halves of the window are gathered with a mask-specific compress network (Hacker's Delight 7-4: six shift-and-merge
    // steps, the steps this mask needs in pext_steps) instead of run by run
    u32 pext_on, pext_n1;            // pext_n1 = sampled bases in the second half (the first half's shift in the key)
    u32 pext_steps[2];               // bit i set: step i (shift by 2^i bits) moves something
    u32 pext_top[2];                 // n when the half's sampled bases are exactly its first n (a plain shift), else 0xFF
    u64 pext_mask[2], pext_mv[2][6];
    // outputs (device)
    u32 *taxon, *missing, *ambig, *n_hits, *hits;
    uint4 *records;     // classify_kernel writes one {taxon, missing, ambig, n_hits} record per unit; unpack_kernel splits it
    u32 *ovf_count;
    u32 *work_counter;  // classify_kernel: next unclaimed unit (wavefronts claim `chunk` units at a time)
    u32 chunk;          // units per claim: classify_chunk(nmates), fewer when the batch is small (a CLI chunk of 112 k reads is 1775 claims of 63: 444 blocks on 256 CUs)
    u64 *ovf_list;
};

}  // namespace bns
