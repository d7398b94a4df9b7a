"""kseq_read (klib/kseq.h:177-225) and bseq_read's loop (kseq_declare.h:106-146) restated character by character in Python:
the checker of the device text parser's tests (tests/test_gpu_ingest.py).  Test infrastructure; pinned to the reference's own
reader by tests/test_ingest_oracle.py (tests/golden/ingest_ref.npz: records kseq_read / bseq_read returned in the build
container).  Slow on purpose: one state machine over the bytes, nothing line-based, so that it shares no idea with the kernels."""

SPACE = b" \t\n\v\f\r"


class KStream:
    def __init__(self, data):
        self.d = bytes(data)
        self.p = 0

    def getc(self):
        if self.p >= len(self.d):
            return -1
        c = self.d[self.p]
        self.p += 1
        return c

    def getuntil(self, sep_line, out, append=False):
        """ks_getuntil2 (klib/kseq.h:96-139): -> (return value, delimiter or None)"""
        if not append:
            del out[:]
        if self.p >= len(self.d):
            return -1, None                                   # !gotany && eof
        i = self.p
        n = len(self.d)
        if sep_line:
            j = self.d.find(b"\n", i)
            j = n if j < 0 else j
        else:
            j = i
            while j < n and self.d[j] not in SPACE:
                j += 1
        out += self.d[i:j]
        dret = self.d[j] if j < n else None
        self.p = j + 1 if j < n else n
        if sep_line and len(out) > 1 and out[-1] == 0x0D:     # :135
            del out[-1]
        return len(out), dret


class KSeq:
    def __init__(self, data):
        self.ks = KStream(data)
        self.last_char = 0
        self.name = bytearray(); self.comment = bytearray(); self.seq = bytearray(); self.qual = bytearray()
        self.start = 0                                        # offset of the record's header byte (for the tests)

    def read(self):
        ks = self.ks
        c = None
        if self.last_char == 0:                               # :183-187
            while True:
                c = ks.getc()
                if c < 0:
                    return c
                if c in (0x3E, 0x40):
                    break
            self.last_char = c
        self.start = ks.p - 1
        del self.comment[:]; del self.seq[:]; del self.qual[:]
        r, d = ks.getuntil(False, self.name)                  # :189
        if r < 0:
            return r
        if d is not None:
            c = d
        if c != 0x0A:                                         # :190 (c keeps its old value when the name ran into the end of the input)
            ks.getuntil(True, self.comment)
        while True:                                           # :195-199
            c = ks.getc()
            if c < 0 or c in (0x3E, 0x2B, 0x40):
                break
            if c == 0x0A:
                continue
            self.seq.append(c)
            ks.getuntil(True, self.seq, append=True)
        self.at_header = c in (0x3E, 0x40)                    # (for reads_cleanly: the next record's header byte is consumed already)
        if c in (0x3E, 0x40):
            self.last_char = c
        if c != 0x2B:
            return len(self.seq)
        while True:                                           # :215
            c = ks.getc()
            if c < 0 or c == 0x0A:
                break
        if c == -1:
            return -2
        while True:                                           # :217
            r, _ = ks.getuntil(True, self.qual, append=True)
            if not (r >= 0 and len(self.qual) < len(self.seq)):
                break
        self.last_char = 0
        if len(self.seq) != len(self.qual):
            return -2
        return len(self.seq)


def trim_readno(name):
    if len(name) > 2 and name[-2] == 0x2F and 0x30 <= name[-1] <= 0x39:
        return name[:-2]
    return name


def read_all(data, data2=None, trim=True, with_pos=False, chunk_size=1 << 20):
    """every record process_dataset's loop sees (classifier.h:306: `while((seqs = bseq_read(chunk_size, &n, ks1, ks2)))` over
    bseq_read, kseq_declare.h:112-146): (name, comment, seq, qual) tuples, mates interleaved.  A kseq_read < 0 ends the CHUNK it
    happens in; the input ends with the first chunk that comes back empty -- so a truncated record in the middle of a chunk is
    dropped and reading goes on behind it, one at the start of a chunk ends everything.  with_pos: a fifth field = offset of the
    header byte"""
    k1 = KSeq(data)
    k2 = KSeq(data2) if data2 is not None else None
    out = []
    while True:
        n = size = 0
        while k1.read() >= 0:
            if k2 is not None and k2.read() < 0:
                break
            for k in (k1, k2):
                if k is None:
                    continue
                nm = bytes(k.name)
                rec = (trim_readno(nm) if trim else nm, bytes(k.comment), bytes(k.seq), bytes(k.qual))
                out.append(rec + (k.start,) if with_pos else rec)
                n += 1
                size += len(k.seq)
            if size >= chunk_size and n % 2 == 0:
                break
        if n == 0:
            break
    return out


def read_until_error(data, trim=True):
    """the records of ONE stream up to the first kseq_read < 0: [(name, comment, seq, qual, header offset)], the return code
    that ended it (-1 end of input, -2 truncated quality) and the offset reading stopped at"""
    k = KSeq(data)
    out = []
    while True:
        rc = k.read()
        if rc < 0:
            return out, rc, k.ks.p
        nm = bytes(k.name)
        out.append((trim_readno(nm) if trim else nm, bytes(k.comment), bytes(k.seq), bytes(k.qual), k.start))


def reads_cleanly(data):
    """True when kseq_read reads `data` to its end without an error AND skips nothing but line ends between records (klib/kseq.h:183-186
    jumps to the next '>' / '@' byte over whatever lies in between: text the device parser hands back as not regular)"""
    k = KSeq(data)
    prev_end = 0
    while True:
        rc = k.read()
        if rc < 0:
            return rc == -1 and not bytes(data[prev_end:]).strip(b"\r\n>@")
        if bytes(data[prev_end:k.start]).strip(b"\r\n"):
            return False
        prev_end = k.ks.p - (1 if k.at_header else 0)
