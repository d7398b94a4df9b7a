"""Random FASTA / FASTQ text for the ingest tests: mostly the regular form (what the device parser takes: wrapped sequences,
blank lines, quality lines that start with '@' / '+', empty sequences, a missing last newline; round 6: CRLF line ends and quality
wrapped over several lines, `crlf` / `wrapq`), and -- with `wild` -- what kseq also reads but the device hands back (text between
records, truncated records, quality of the wrong length)."""
import numpy as np

ALPHA = np.frombuffer(b"ACGTACGTACGTACGTNacgtnRYK", dtype=np.uint8)
QUAL = np.frombuffer(b"!#+5:@@@IIIIFFFF+++~", dtype=np.uint8)


def rand_seq(rng, n):
    return ALPHA[rng.integers(0, ALPHA.size, size=n)].tobytes()


def rand_qual(rng, n):
    return QUAL[rng.integers(0, QUAL.size, size=n)].tobytes()


def wrap(b, w):
    return [b[i:i + w] for i in range(0, len(b), w)] if b else []


def make_doc(rng, n_records, wild=0.0, kinds=("fastq", "fasta"), max_len=400, final_newline=True, fastq_comments=True, crlf=None, wrapq=None):
    """fastq_comments=False: no comment on a record that has a quality line (the reference's kseq2bseq1, kseq_declare.h:54-73,
    allocates one byte too few for a record with BOTH: the live cross-check against the compiled reference keeps clear of it)"""
    # (round 6: CRLF records and wrapped quality are part of the regular form; `wild` still mixes them in at its old rates)
    crlf = wild * 0.2 if crlf is None else crlf
    wrapq = wild * 0.25 if wrapq is None else wrapq
    out = []
    for i in range(n_records):
        kind = kinds[int(rng.integers(0, len(kinds)))]
        L = int(rng.integers(0, max_len)) if rng.random() < 0.9 else int(rng.integers(0, 5))
        seq = rand_seq(rng, L)
        name = b"r%d" % i
        r = rng.random()
        if r < 0.2:
            name += b"/%d" % int(rng.integers(1, 3))
        elif r < 0.3:
            name = b""
        hdr = (b"@" if kind == "fastq" or rng.random() < 0.1 else b">") + name
        r = rng.random()
        if kind == "fastq" and not fastq_comments:
            r = 1.0
        if r < 0.3:
            hdr += b" a comment with words"
        elif r < 0.4:
            hdr += b"\tx"
        lines = [hdr]
        r = rng.random()
        if r < 0.6:
            sl = [seq] if seq else ([b""] if rng.random() < 0.5 else [])
        else:
            sl = wrap(seq, int(rng.integers(1, 90)))
        if rng.random() < 0.1:                              # blank lines inside the sequence
            k = int(rng.integers(0, len(sl) + 1))
            sl = sl[:k] + [b""] + sl[k:]
        lines += sl
        if kind == "fastq":
            lines.append(b"+" + (name if rng.random() < 0.2 else b""))
            q = bytearray(rand_qual(rng, L))
            if L and rng.random() < 0.3:
                q[0] = ord("@") if rng.random() < 0.6 else ord("+")
            q = bytes(q)
            w = rng.random()
            if rng.random() < wrapq and L > 4:
                lines += wrap(q, int(rng.integers(2, L)))   # wrapped quality
            elif w < wild * 0.10 and L > 2:
                lines.append(q[:-1])                        # short quality (kseq reads on into the next record)
            elif w < wild * 0.20:
                lines.append(q + b"I")                      # long quality: error -2
            else:
                lines.append(q)
        if rng.random() < 0.1:
            lines.append(b"")
        if rng.random() < wild * 0.15:
            lines.append(b"stray text between records")
        nl = b"\r\n" if rng.random() < crlf else b"\n"
        out.append(nl.join(lines) + nl)
    doc = b"".join(out)
    if not final_newline and doc.endswith(b"\n"):
        doc = doc[:-1]
    return doc
