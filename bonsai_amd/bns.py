"""Python surface of the reference's pybind11 module (python/bns.cpp) over the GPU encoder.

Same function names, argument names and defaults as the reference module; every function returns numpy uint64 arrays of
what Encoder<score::Lex>::for_each emits (python/bns.cpp:87-129, 175-199), or, for the `*_r` / rolling variants, what
RollingHasher<uint64_t>::for_each_hash emits (python/bns.cpp:42-86, 151-174) -- with character tables from a restated
generator, since the reference's come from an un-vendored one (self-consistent, reference-unverified).
"""
import numpy as np

from . import _lib, hostio
from .context import Context, concat_reads

_ctx = {}


def _context(device=0):
    if device not in _ctx:
        _ctx[device] = Context(device)
    return _ctx[device]


def _configure(ctx, k, spacing, w, canon, path_overload):
    gaps = hostio.parse_spacing(spacing, k) if spacing else None
    spaced = gaps is not None and bool(np.any(gaps))
    # the string overload of Encoder::for_each emits nothing for a spaced seed (encoder.h:437-440); the path
    # overloads reach for_each_uncanon_spaced (encoder.h:453-463)
    ctx.set_encoder(k, gaps, canonicalize=canon, spaced_intended=path_overload)
    comb = k + (int(gaps.sum()) if gaps is not None else 0)
    if w and w > comb:
        ctx.set_window(w, _lib.SCORE_LEX)           # Encoder<> = score::Lex (parity unpinned)


def _encode(ctx, seqs, unique):
    bases, offsets = concat_reads(seqs)
    out = ctx.encode(bases, offsets)
    if unique:
        out = [np.unique(a) for a in out]           # python/bns.cpp:16-26: set, then sorted
    return out


def from_str(str, k=31, spacing=None, w=0, canon=True, reserve=1024, device=0):  # noqa: A002 (reference argument name)
    """python/bns.cpp:112-129"""
    ctx = _context(device)
    _configure(ctx, k, spacing, w, canon, path_overload=False)
    return _encode(ctx, [str], False)[0]


def _records(path):
    recs, _ = hostio.read_fastx(path)
    return recs


def from_fasta(path, k=31, spacing="", w=0, canon=True, reserve=1024, unique=False, device=0):
    """python/bns.cpp:106-111: every record of the file, concatenated in order"""
    ctx = _context(device)
    _configure(ctx, k, spacing, w, canon, path_overload=True)
    parts = _encode(ctx, [r[2] for r in _records(path)], False)
    allk = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint64)
    return np.unique(allk) if unique else allk


def seqlist(path, k=31, spacing="", w=0, canon=True, reserve=1024, unique=False, rolling=False, device=0):
    """python/bns.cpp:87-105: one array per record (string overload per record)"""
    ctx = _context(device)
    if rolling:                                     # python/bns.cpp:90,96-97: RollingHasher<uint64_t>(k), i.e. not canonical
        return _rolling(ctx, [r[2] for r in _records(path)], k, False, unique)
    _configure(ctx, k, spacing, w, canon, path_overload=False)
    return _encode(ctx, [r[2] for r in _records(path)], unique)


def seqdict(path, k=31, spacing="", w=0, device=0):
    """python/bns.cpp:175-199: record name -> k-mers.  (The reference re-encodes the whole FILE for every record, an
    obvious slip; this returns each record's own k-mers.)"""
    ctx = _context(device)
    _configure(ctx, k, spacing, w, True, path_overload=True)
    recs = _records(path)
    return {r[0].decode(): a for r, a in zip(recs, _encode(ctx, [r[2] for r in recs], False))}


def _rolling(ctx, seqs, k, canon, unique):
    bases, offsets = concat_reads(seqs)
    out = ctx.rolling_hash(bases, offsets, k, canon)
    return [np.unique(a) for a in out] if unique else out


def from_fasta_r(path=None, k=31, canon=True, reserve=1024, unique=False, device=0):
    """python/bns.cpp:82-86: RollingHasher<uint64_t>(k, canon) over every record of the file, concatenated in order"""
    parts = _rolling(_context(device), [r[2] for r in _records(path)], k, canon, False)
    allh = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint64)
    return np.unique(allh) if unique else allh


def seqdict_r(path, k=31, device=0):
    """python/bns.cpp:151-174: record name -> rolling hashes, RollingHasher<uint64_t>(k) (not canonical).  (Like seqdict, the
    reference hashes the whole FILE for every record; this returns each record's own values.)"""
    recs = _records(path)
    return {r[0].decode(): a for r, a in zip(recs, _rolling(_context(device), [r[2] for r in recs], k, False, False))}
