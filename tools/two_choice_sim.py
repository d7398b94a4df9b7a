#!/usr/bin/env python
"""Round 6: would WHOLE-GROUP TWO-CHOICE placement help the crowded every-k-mer table (8e9 keys, 46 % of the slots: 6.4 probe passes
and 39.7 bucket fetches per 150-bp read, 0.20 of the roofline -- DESIGN §9.2)?  A geometry simulation, no GPU: every k-mer of a random
genome (k = 31, minimizer window 8 as the loader picks for that db), buckets of 10 keys, the table at a given load.

  one-choice   what the loader builds today (docs/TABLE_LAYOUT.md "Round 4"): a minimizer GROUP's home bucket is hash(minimizer); a
               bucket keeps the groups that fit whole, largest first (by 3-bit tag); the other keys take what room is left at home,
               then the next three buckets, then the overflow table.  A lookup leaves its home bucket only when its tag bit is set.
  two-choice   every group has TWO candidate homes h1, h2 (two hashes of the minimizer); groups are placed whole, largest first, in
               h1 when it fits there, else in h2 when it fits there (a header bit per tag in h1 says "this group lives in its second
               home"), else split over h1, h2, their chains, the overflow table.  A lookup probes h1, then h2, then the chains.

Per read (150 bp = 120 k-mers = rounds of 64 + 56 lookups, a round's probe passes = the DEEPEST lookup of the round, as in
classify_kernel): probe passes, distinct bucket fetches, share of rounds that end in the overflow table; per key: share outside the
first bucket probed.  usage: two_choice_sim.py [log2_buckets=20] [load=0.46] [reads=20000]"""
import sys
import numpy as np

K, SPAN, CAP, CHAIN = 31, 8, 10, 4
M = K - SPAN


def mix(x, c):
    x = (x ^ (x >> np.uint64(33))) * np.uint64(c)
    x = (x ^ (x >> np.uint64(29))) * np.uint64(0xC4CEB9FE1A85EC53)
    return x ^ (x >> np.uint64(32))


def kmers_of(codes, k):
    """forward k-mers (2 bits per base, MSB first) of a code array -> uint64[n - k + 1]"""
    n = codes.size - k + 1
    v = np.zeros(n, dtype=np.uint64)
    for i in range(k):
        v = (v << np.uint64(2)) | codes[i:i + n].astype(np.uint64)
    return v


def groups_of(genome):
    """-> per k-mer: its key, the minimizer value (the smallest hash among its SPAN + 1 m-mers decides; the m-mer is the identity)"""
    mm = kmers_of(genome, M)
    hm = mix(mm, 0xFF51AFD7ED558CCD)
    n = genome.size - K + 1
    best = hm[:n].copy(); arg = np.zeros(n, dtype=np.int64)
    for j in range(1, SPAN + 1):
        c = hm[j:j + n]
        better = c < best
        best = np.where(better, c, best); arg = np.where(better, j, arg)
    ident = mm[np.arange(n) + arg]
    return kmers_of(genome, K), ident


def simulate(log2_buckets, load, n_reads, seed=1):
    rng = np.random.default_rng(seed)
    NB = 1 << log2_buckets
    n_keys_target = int(load * CAP * NB)
    genome = rng.integers(0, 4, size=n_keys_target + K - 1, dtype=np.uint8)
    key, ident = groups_of(genome)
    # distinct keys only (a random genome of this size has almost no repeats; keep the first occurrence)
    _, first = np.unique(key, return_index=True)
    keep = np.zeros(key.size, dtype=bool); keep[first] = True
    n_keys = int(keep.sum())
    h1 = (mix(ident, 0x9E3779B97F4A7C15) % np.uint64(NB)).astype(np.int64)
    h2 = (mix(ident, 0xD6E8FEB86659FD93) % np.uint64(NB)).astype(np.int64)
    tag = (mix(ident, 0xA0761D6478BD642F) & np.uint64(7)).astype(np.int64)
    out = {}
    for policy in ("one-choice", "two-choice"):
        room = np.full(NB + CHAIN, CAP, dtype=np.int64)
        depth = np.zeros(key.size, dtype=np.int64)             # probes until the key is found (1 = first bucket probed)
        in_overflow = np.zeros(key.size, dtype=bool)
        idx = np.flatnonzero(keep)
        if policy == "one-choice":
            # groups by (home, tag); whole groups largest first per home
            gid = h1[idx] * 8 + tag[idx]
            order = np.argsort(gid, kind="stable")
            gs, start, cnt = np.unique(gid[order], return_index=True, return_counts=True)
            home = gs // 8
            resident = np.zeros(gs.size, dtype=bool)
            # per home: sort its groups by size desc and keep while they fit
            o2 = np.lexsort((-cnt, home))
            used = np.zeros(NB + CHAIN, dtype=np.int64)
            for g in o2:                                        # (python loop over ~NB * 2 groups)
                hb = home[g]
                if used[hb] + cnt[g] <= CAP:
                    used[hb] += cnt[g]; resident[g] = True
            room[:NB] -= used[:NB]
            res_key = np.repeat(resident, cnt)
            kidx = idx[order]
            depth[kidx[res_key]] = 1
            spill = kidx[~res_key]
            spill = spill[rng.permutation(spill.size)]           # arrival order
            for i in spill:
                hb = h1[i]
                for d in range(CHAIN):
                    if room[hb + d] > 0:
                        room[hb + d] -= 1; depth[i] = d + 1
                        break
                else:
                    depth[i] = CHAIN + 1; in_overflow[i] = True
            # a lookup of a key whose group is whole at home ends at 1; a spilled key that found room AT HOME still has depth 1
        else:
            gid = ident[idx]
            order = np.argsort(gid, kind="stable")
            gs, start, cnt = np.unique(gid[order], return_index=True, return_counts=True)
            kidx = idx[order]
            g_h1 = h1[kidx[start]]; g_h2 = h2[kidx[start]]
            big_first = np.argsort(-cnt, kind="stable")
            place = np.zeros(gs.size, dtype=np.int64)            # 1: whole in h1, 2: whole in h2, 0: split
            for g in big_first:
                c = cnt[g]
                if room[g_h1[g]] >= c:
                    room[g_h1[g]] -= c; place[g] = 1
                elif room[g_h2[g]] >= c:
                    room[g_h2[g]] -= c; place[g] = 2
            pk = np.repeat(place, cnt)
            depth[kidx[pk == 1]] = 1
            depth[kidx[pk == 2]] = 2
            split = kidx[pk == 0]
            split = split[rng.permutation(split.size)]
            for i in split:
                # probe order of a lookup: h1, h2, then the chains of h1 and of h2 in turn
                seq = [h1[i], h2[i]] + [b for d in range(1, CHAIN) for b in (h1[i] + d, h2[i] + d)]
                for p, b in enumerate(seq):
                    if room[b] > 0:
                        room[b] -= 1; depth[i] = p + 1
                        break
                else:
                    depth[i] = len(seq) + 1; in_overflow[i] = True
        # a duplicate k-mer finds its first occurrence's entry
        first_of = np.zeros(key.size, dtype=np.int64)
        srt = np.argsort(key, kind="stable")
        ks = key[srt]
        grp_start = np.r_[True, ks[1:] != ks[:-1]]
        lead = np.maximum.accumulate(np.where(grp_start, np.arange(key.size), 0))
        first_of[srt] = srt[lead]
        depth_all = depth[first_of]; ovf_all = in_overflow[first_of]
        # reads
        starts = rng.integers(0, genome.size - 150, size=n_reads)
        passes = fetches = ovf_rounds = rounds = 0
        for s in starts:
            d = depth_all[s:s + 120]; o = ovf_all[s:s + 120]
            hb1 = h1[s:s + 120]; hb2 = h2[s:s + 120]
            for a, b in ((0, 64), (64, 120)):
                dd = d[a:b]
                passes += int(dd.max()); rounds += 1
                ovf_rounds += int(o[a:b].any())
                # distinct buckets fetched: pass p fetches, for every lookup still walking, the p-th bucket of its probe order
                seen = 0
                for p in range(1, int(dd.max()) + 1):
                    walking = dd >= p
                    if policy == "one-choice":
                        bk = hb1[a:b][walking] + (p - 1)
                    else:
                        base = np.where(p % 2 == 1, hb1[a:b][walking], hb2[a:b][walking]) if p <= 2 else np.where(p % 2 == 1, hb1[a:b][walking], hb2[a:b][walking])
                        bk = base + (0 if p <= 2 else (p - 1) // 2)
                    seen += np.unique(bk).size
                fetches += seen
        off_home = float((depth[idx] > 1).mean())
        out[policy] = {"keys": n_keys, "load": n_keys / float(CAP * NB), "off_first_bucket": off_home, "in_overflow": float(in_overflow[idx].mean()),
                       "passes_per_read": passes / float(n_reads), "fetches_per_read": fetches / float(n_reads),
                       "rounds_ending_in_overflow": ovf_rounds / float(rounds)}
    return out


def main():
    lb = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    load = float(sys.argv[2]) if len(sys.argv) > 2 else 0.46
    n_reads = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
    print("# whole-group two-choice placement against today's one-choice group-aware fill: 2^%d buckets of %d keys, every k-mer of a random genome, k = %d, window %d" % (lb, CAP, K, SPAN))
    for ld in ([load] if len(sys.argv) > 2 else [0.34, 0.46, 0.49]):
        r = simulate(lb, ld, n_reads)
        for pol, v in r.items():
            print("load %.2f  %-10s  keys outside the first bucket probed %.3f, in the overflow table %.4f;  per read: %.2f probe passes, %.1f bucket fetches, %.3f of the rounds end in the overflow table"
                  % (v["load"], pol, v["off_first_bucket"], v["in_overflow"], v["passes_per_read"], v["fetches_per_read"], v["rounds_ending_in_overflow"]))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
