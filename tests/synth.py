"""Seeded synthetic fixtures: taxonomy, genomes with shared segments (so LCAs at species / genus /
root level occur in the db), simulated reads.  Test infrastructure only."""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)

# child -> parent; root is 1.  Leaves (strains): 1001,1002 (species 101) 1003 (species 102) [genus 11]
#                                                1004 (species 111) [genus 12]        [phylum 2]
#                                                2001,2002 (species 201) [genus 21]   [phylum 3]
TAX_PAIRS = [(1, 1), (2, 1), (3, 1), (11, 2), (12, 2), (21, 3), (101, 11), (102, 11), (111, 12), (201, 21),
             (1001, 101), (1002, 101), (1003, 102), (1004, 111), (2001, 201), (2002, 201)]
LEAVES = [1001, 1002, 1003, 1004, 2001, 2002]


def rand_seq(rng, n):
    return ACGT[rng.integers(0, 4, size=n)]


def parent_of():
    return {c: p for c, p in TAX_PAIRS if c != 1}


def ancestors(t):
    par = parent_of()
    out = [t]
    while t in par:
        t = par[t]
        out.append(t)
    return out


def make_genomes(rng, genome_len=6000, seg=500):
    """Each leaf genome is a chain of segments; a segment is private, or shared with every leaf under
    one of the leaf's ancestors (species / genus / phylum / root)."""
    pools = {}
    genomes = {}
    for leaf in LEAVES:
        anc = ancestors(leaf)          # leaf, species, genus, phylum, root
        parts = []
        for s in range(genome_len // seg):
            level = rng.choice([0, 0, 0, 1, 2, 3, 4])       # mostly private
            if level == 0:
                parts.append(rand_seq(rng, seg))
            else:
                key = (anc[min(level, len(anc) - 1)], s % 3)
                if key not in pools:
                    pools[key] = rand_seq(rng, seg)
                parts.append(pools[key])
        genomes[leaf] = np.concatenate(parts)
    return genomes


def mutate(rng, seq, sub_rate=0.01, n_rate=0.001, lower_rate=0.0):
    seq = seq.copy()
    m = rng.random(seq.size) < sub_rate
    seq[m] = ACGT[rng.integers(0, 4, size=int(m.sum()))]
    m = rng.random(seq.size) < n_rate
    seq[m] = ord("N")
    if lower_rate:
        m = rng.random(seq.size) < lower_rate
        seq[m] |= 0x20
    return seq


COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    COMP[a] = b


def revcomp(seq):
    return COMP[seq[::-1]]


def simulate_reads(rng, genomes, n, length=150, sub_rate=0.01, n_rate=0.001, random_frac=0.05, var_len=False,
                   lower_rate=0.0):
    """Returns list of uint8 arrays."""
    leaves = list(genomes)
    out = []
    for _ in range(n):
        L = int(rng.integers(20, 2 * length)) if var_len else length
        if rng.random() < random_frac:
            r = rand_seq(rng, L)
        else:
            g = genomes[leaves[int(rng.integers(len(leaves)))]]
            L = min(L, g.size)
            st = int(rng.integers(0, g.size - L + 1))
            r = g[st:st + L]
            if rng.random() < 0.5:
                r = revcomp(r)
            r = mutate(rng, r, sub_rate, n_rate, lower_rate)
        out.append(np.ascontiguousarray(r))
    return out


def concat(reads):
    offsets = np.zeros(len(reads) + 1, dtype=np.uint64)
    if reads:
        offsets[1:] = np.cumsum([r.size for r in reads], dtype=np.uint64)
        bases = np.concatenate(reads) if offsets[-1] else np.zeros(0, dtype=np.uint8)
    else:
        bases = np.zeros(0, dtype=np.uint8)
    return np.ascontiguousarray(bases, dtype=np.uint8), offsets


class World:
    pass


def make_world(oracle, seed=11, k=31, genome_len=6000, gaps=None, canon=True):
    rng = np.random.default_rng(seed)
    w = World()
    w.k, w.gaps, w.canon = k, gaps, canon
    w.tax = oracle.Taxonomy(pairs=TAX_PAIRS)
    w.parent = w.tax.parent
    w.genomes = make_genomes(rng, genome_len)
    w.table = oracle.Table()
    for leaf, g in w.genomes.items():
        oracle.lca_map_add(w.table, w.tax, k, g.tobytes(), leaf, gaps=gaps, canon=canon)
    w.flags, w.keys, w.vals = w.table.arrays()
    w.n_buckets = w.table.n_buckets
    w.rng = rng
    return w


def write_nodes_dmp(path, pairs=TAX_PAIRS):
    with open(path, "w") as f:
        for c, p in pairs:
            f.write("%d\t|\t%d\t|\tno rank\t|\t\t|\n" % (c, p))


def write_bgzf(path, data, block=65280, level=6, member_sizes=None):
    """`data` as a BGZF file (SAM spec 4.1: gzip members of <= 64 KiB of text, compressed size in a 'BC' extra subfield, an
    empty member at the end).  member_sizes: an iterable of text sizes to cut the members by (cycled), instead of `block`."""
    import itertools
    import struct
    import zlib
    sizes = itertools.cycle(member_sizes) if member_sizes else itertools.repeat(block)

    def member(chunk):
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = co.compress(chunk) + co.flush()
        bsize = 12 + 6 + len(body) + 8 - 1
        assert bsize < 65536
        return (b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize) + body
                + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    with open(path, "wb") as f:
        at = 0
        while at < len(data):
            n = min(next(sizes), 65280)
            f.write(member(data[at:at + n]))
            at += n
        f.write(member(b""))
