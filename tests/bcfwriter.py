"""Test helper: a minimal BCF2.2 (+ CSI index) writer following the VCF/BCF specification
(hts-specs VCFv4.3 section 6) -- there is no htslib / bcftools in this image, so the fixtures for the
C++ BCF reader are produced here.  Only what the reader consumes is written: CHROM, POS, ID, alleles,
FILTER, FORMAT/GT (as int8, int16 or int32 vectors, with end-of-vector padding for mixed ploidy) and FORMAT/DS
(float32 vectors, missing = 0x7F800001, end of vector = 0x7F800002).
"""
import struct
import zlib

import numpy as np

INT8_END, INT16_END, INT32_END = -127, -32767, -2147483647
LEVEL = 6  # deflate level of the BGZF blocks (bench.py lowers it: the fixture is written once per run)


def bgzf_block(data: bytes) -> bytes:
    c = zlib.compressobj(LEVEL, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    bsize = len(comp) + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize)
            + comp + struct.pack("<II", zlib.crc32(data), len(data)))


def typed_int(v: int) -> bytes:
    if -120 <= v <= 127:
        return b"\x11" + struct.pack("<b", v)
    if -32760 <= v <= 32767:
        return b"\x12" + struct.pack("<h", v)
    return b"\x13" + struct.pack("<i", v)


def typed_desc(n: int, t: int) -> bytes:
    if n < 15:
        return bytes([(n << 4) | t])
    return bytes([0xF0 | t]) + typed_int(n)


def typed_str(s: str) -> bytes:
    b = s.encode()
    if not b:
        return b"\x07"
    return typed_desc(len(b), 7) + b


def reg2bin(beg: int, end: int, min_shift: int = 14, depth: int = 5) -> int:
    """hts-specs CSIv1: bin of the half-open 0-based interval [beg, end)"""
    l, s, t = depth, min_shift, ((1 << depth * 3) - 1) // 7
    end -= 1
    while l > 0:
        if beg >> s == end >> s:
            return t + (beg >> s)
        l -= 1
        s += 3
        t -= 1 << (l * 3)
    return 0


def write_bcf(path, contigs, samples, records, gt_dtype=np.int8, filters=("PASS", "FAIL"),
              block_bytes=0xff00, with_csi=True, extra_header=()):
    """records: dicts with contig, pos (1-based), id, ref, alts (list), filters (list of names; [] = '.'),
    gts (int array [n_samples, ploidy] in the bcf GT encoding, padded with the int32 vector end)."""
    hdr = ["##fileformat=VCFv4.2"]
    ids = []
    for f in filters:
        hdr.append('##FILTER=<ID=%s,Description="%s">' % (f, "All filters passed" if f == "PASS" else f))
        ids.append(f)
    if "PASS" not in ids:
        ids.insert(0, "PASS")
    for h in extra_header:          # further FILTER/INFO/FORMAT lines take dictionary slots too
        hdr.append(h)
        if h.startswith(("##FILTER=", "##INFO=", "##FORMAT=")):
            name = h.split("ID=", 1)[1].split(",", 1)[0].rstrip(">")
            if name not in ids:
                ids.append(name)
    for c in contigs:
        hdr.append("##contig=<ID=%s>" % c)
    hdr.append('##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">')
    ids.append("GT")
    if any(r.get("ds") is not None for r in records):
        hdr.append('##FORMAT=<ID=DS,Number=A,Type=Float,Description="ALT allele dosage">')
        ids.append("DS")
    hdr.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(samples))
    text = ("\n".join(hdr) + "\n").encode() + b"\0"
    gt_key = ids.index("GT")
    ttype = {np.int8: 1, np.int16: 2, np.int32: 3}[gt_dtype]
    vend = {np.int8: INT8_END, np.int16: INT16_END, np.int32: INT32_END}[gt_dtype]

    out = bytearray()
    buf = bytearray()

    def flush():
        nonlocal buf
        if buf:
            out.extend(bgzf_block(bytes(buf)))
            buf = bytearray()

    def add(data: bytes):
        nonlocal buf
        start = (len(out) << 16) | len(buf)
        i = 0
        while i < len(data):
            room = block_bytes - len(buf)
            if room == 0:
                flush()
                room = block_bytes
            buf.extend(data[i:i + room])
            i += room
        if len(buf) >= block_bytes:
            flush()
        return start, (len(out) << 16) | len(buf)

    add(b"BCF\2\2" + struct.pack("<I", len(text)) + text)
    index = [dict() for _ in contigs]
    for r in records:
        chrom = contigs.index(r["contig"])
        pos0 = r["pos"] - 1
        rlen = len(r["ref"])
        alleles = [r["ref"]] + list(r["alts"])
        shared = struct.pack("<iiiI", chrom, pos0, rlen, 0x7F800001)
        has_gt = r.get("gts") is not None
        g = np.asarray(r["gts"], dtype=np.int64) if has_gt else np.zeros((0, 0), np.int64)
        ns, ploidy = (g.shape if g.ndim == 2 else (len(samples), 0))
        ds = r.get("ds")
        n_fmt = (1 if ns else 0) + (1 if ds is not None else 0)
        shared += struct.pack("<II", (len(alleles) << 16) | 0, (n_fmt << 24) | len(samples))
        shared += typed_str(r.get("id", ".") if r.get("id", ".") != "." else "")
        for a in alleles:
            shared += typed_str(a)
        fl = r.get("filters", [])
        if fl:
            shared += typed_desc(len(fl), 1) + bytes(ids.index(f) for f in fl)
        else:
            shared += b"\x00"
        indiv = b""
        if ns:
            g = np.where(g == INT32_END, vend, g).astype(gt_dtype)
            indiv = typed_int(gt_key) + typed_desc(ploidy, ttype) + g.tobytes()
        if ds is not None:   # float vector: NaN -> the BCF missing value; 0x7F800002 payloads (end of vector) stay
            d = np.ascontiguousarray(ds, dtype=np.float32).reshape(len(samples), -1)
            bits = d.view(np.uint32).copy()
            bits[np.isnan(d) & (bits != 0x7F800002)] = 0x7F800001
            indiv += typed_int(ids.index("DS")) + typed_desc(d.shape[1], 5) + bits.tobytes()
        vs, ve = add(struct.pack("<II", len(shared), len(indiv)) + shared + indiv)
        index[chrom].setdefault(reg2bin(pos0, pos0 + rlen), []).append((vs, ve))
    flush()
    out.extend(bgzf_block(b""))
    open(path, "wb").write(bytes(out))
    if not with_csi:
        return
    t = bytearray(b"CSI\1" + struct.pack("<iii", 14, 5, 0) + struct.pack("<i", len(contigs)))
    for bins in index:
        t += struct.pack("<i", len(bins))
        for b, chunks in bins.items():
            t += struct.pack("<IQi", b, chunks[0][0], 1) + struct.pack("<QQ", chunks[0][0], chunks[-1][1])
    t += struct.pack("<Q", 0)
    tb = bytearray()
    for i in range(0, len(t), 0xff00):
        tb += bgzf_block(bytes(t[i:i + 0xff00]))
    tb += bgzf_block(b"")
    open(path + ".csi", "wb").write(bytes(tb))


def records_from_oracle_vcf(vcf, filter_names=("PASS", "FAIL")):
    """oracle.refcpu.Vcf -> (contigs, records) for write_bcf"""
    contigs, recs = [], []
    for r in vcf.records:
        if r.contig not in contigs:
            contigs.append(r.contig)
        fl = [] if r.filt == "." else r.filt.split(";")
        recs.append(dict(contig=r.contig, pos=r.pos, id=".", ref=r.ref, alts=r.alts, filters=fl,
                         gts=np.asarray(r.gts).reshape(len(vcf.samples), r.ploidy)))
    return contigs, recs
