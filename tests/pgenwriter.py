"""Writer of the ONE .pgen flavour the host reads: PLINK 2 storage mode 0x02 (fixed-width hard calls), with its
.pvar and .psam.  Test infrastructure, the twin of tests/bcfwriter.py.

PARITY UNPINNED: neither plink2 nor the .pgen specification is in this image and the reference (which reads VCF/BCF
only) holds no .pgen fixture, so the reader in nimpress_host.cpp and this writer vouch for each other -- exactly as the
BCF2/CSI reader and tests/bcfwriter.py do.  Layout as published for pgenlib: bytes 0-1 magic 0x6c 0x1b, byte 2 storage
mode 0x02, uint32 variant count, uint32 sample count, one flag byte (0x40: every ALT allele "trusted"), then per
variant ceil(N/4) bytes: sample i in bits 2(i mod 4) of byte i div 4, code = number of ALT alleles, 3 = missing."""
import numpy as np


def pack_rows(alt_counts: np.ndarray, missing: np.ndarray) -> bytes:
    """alt_counts [m, n] in {0, 1, 2}, missing [m, n] bool -> the fixed-width records"""
    m, n = alt_counts.shape
    code = np.where(missing, 3, alt_counts).astype(np.uint8)
    pad = (-n) % 4
    if pad:
        code = np.concatenate([code, np.zeros((m, pad), np.uint8)], axis=1)
    code = code.reshape(m, -1, 4)
    return (code[:, :, 0] | (code[:, :, 1] << 2) | (code[:, :, 2] << 4) | (code[:, :, 3] << 6)).astype(np.uint8).tobytes()


def write_pgen(prefix: str, samples, variants, alt_counts: np.ndarray, missing: np.ndarray, pvar_header: bool = True,
               psam_fid: bool = False):
    """variants: list of (contig, pos, id, ref, alt).  Writes prefix.pgen / .pvar / .psam."""
    m, n = alt_counts.shape
    assert m == len(variants) and n == len(samples)
    with open(prefix + ".pgen", "wb") as f:
        f.write(b"\x6c\x1b\x02")
        f.write(np.uint32(m).tobytes())
        f.write(np.uint32(n).tobytes())
        f.write(b"\x40")
        f.write(pack_rows(alt_counts, missing))
    with open(prefix + ".pvar", "w") as f:
        if pvar_header:
            f.write("##fileformat=PVARv1.0\n#CHROM\tPOS\tID\tREF\tALT\n")
            for c, p, i, r, a in variants:
                f.write("%s\t%d\t%s\t%s\t%s\n" % (c, p, i, r, a))
        else:   # headerless = the .bim layout: chrom, id, cM, pos, ALT, REF
            for c, p, i, r, a in variants:
                f.write("%s\t%s\t0\t%d\t%s\t%s\n" % (c, i, p, a, r))
    with open(prefix + ".psam", "w") as f:
        if psam_fid:
            f.write("#FID\tIID\tSEX\n")
            for k, s in enumerate(samples):
                f.write("F%d\t%s\tNA\n" % (k, s))
        else:
            f.write("#IID\tSEX\n")
            for s in samples:
                f.write("%s\tNA\n" % s)
