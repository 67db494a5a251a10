"""CPU: the arithmetic behind block mode (sparkzstd_amd/csrc/mzd_exec_blk.hip), restated in numpy -- the position patterns that
stand in for the data before a block's start, and the way k_blk_fixup reads a byte's origin back from the passes.
(The kernels themselves are covered by the -m gpu parity tests; this pins the scheme they implement.)"""
import numpy as np
import pytest


def pattern(x, p, n_passes=4):
    """k_blk_pattern: what pass p puts in place of the byte at frame-relative position x: byte 0, byte 1, byte 0 XOR (1 + bits
    16..22), bits 23..30 (the last one only for frames whose matches reach back 8 MiB or more)"""
    x = np.asarray(x, dtype=np.uint64)
    if p < 2:
        return ((x >> np.uint64(8 * p)) & np.uint64(0xFF)).astype(np.uint8)
    if p == 2:
        return ((x & np.uint64(0xFF)) ^ (((x >> np.uint64(16)) & np.uint64(0x7F)) + np.uint64(1))).astype(np.uint8)
    return ((x >> np.uint64(23)) & np.uint64(0xFF)).astype(np.uint8)


def origin(planes, n_passes, S=None, high=True):
    """fix_origin: (derived?, origin) of a byte from what the passes made of it.  n_passes 3: frames below 8 MiB; 4 with `high`:
    the fourth plane holds bits 23..30; 4 without: the origin is the position in [S - 8 MiB, S) with the 23 bits of passes 0-2"""
    a, e = planes[0].astype(np.uint64), planes[2].astype(np.uint64)
    d = a ^ e
    low = a | (planes[1].astype(np.uint64) << np.uint64(8)) | (((d - np.uint64(1)) & np.uint64(0xFF)) << np.uint64(16))
    if n_passes == 3:
        org = low
    elif high:
        org = low | (planes[3].astype(np.uint64) << np.uint64(23))
    else:
        s1 = np.uint64(S - 1)
        org = s1 - ((s1 - low) & np.uint64(0x7FFFFF))
    return d != 0, org.astype(np.uint32)


@pytest.mark.parametrize("n_passes,limit", [(3, 1 << 23), (4, (1 << 31) - 65536)])
def test_every_position_is_told_apart_from_a_literal_and_read_back(n_passes, limit):
    rng = np.random.default_rng(5)
    x = np.concatenate([np.arange(0, 70000, dtype=np.uint64), rng.integers(0, limit, size=400000, dtype=np.uint64),
                        np.array([limit - 1, limit - 255, limit - 256, limit - 257], dtype=np.uint64)])
    planes = [pattern(x, p, n_passes) for p in range(n_passes)]
    derived, org = origin(planes, n_passes)
    assert derived.all()  # pass 0 and pass 2 never agree on a copied position
    assert (org.astype(np.uint64) == x).all()
    # a byte that does not derive from earlier blocks is the same in every pass: never taken for a derived one
    lit = rng.integers(0, 256, size=1000, dtype=np.uint8)
    derived, _ = origin([lit] * n_passes, n_passes)
    assert not derived.any()


@pytest.mark.parametrize("S", [1, 4097, (1 << 23) - 1, 1 << 23, (1 << 23) + 5, (1 << 27) + 12345, (1 << 31) - 65537])
def test_three_passes_name_the_origin_when_matches_stay_within_8_mib(S):
    """Without the pass of the high bits: every position of [S - 8 MiB, S) (clipped at 0) is read back from its low 23 bits and
    the segment's start -- what k_blk_fixup does for a frame whose largest offset k_seq_q4 reported below 8 MiB."""
    rng = np.random.default_rng(S & 0xFFFF)
    lo = max(0, S - (1 << 23))
    x = np.unique(np.concatenate([rng.integers(lo, S, size=200000, dtype=np.uint64),
                                  np.array([lo, S - 1, (lo + S) // 2], dtype=np.uint64)]))
    planes = [pattern(x, p) for p in range(3)]
    derived, org = origin(planes, 4, S=S, high=False)
    assert derived.all() and (org.astype(np.uint64) == x).all()


def test_fixup_walk_on_a_toy_frame():
    """Three 'blocks' whose bytes are literals or copies of earlier positions (possibly of copies): executing every block
    against the patterns and then gathering block after block gives what executing them in order gives."""
    rng = np.random.default_rng(9)
    n_blocks, bs, n_passes = 6, 4096, 3
    n = n_blocks * bs
    src = np.full(n, -1, dtype=np.int64)  # -1: literal, else the position the byte is copied from
    for i in range(n):
        if i > 64 and rng.random() < 0.7:
            src[i] = i - int(rng.integers(1, min(i, 3 * bs)))
    lit = rng.integers(0, 256, size=n, dtype=np.uint8)
    want = lit.copy()
    for i in range(n):
        if src[i] >= 0:
            want[i] = want[src[i]]
    # per block and pass: bytes before the block's start read as the pass's pattern
    planes = [np.zeros(n, dtype=np.uint8) for _ in range(n_passes)]
    for b in range(n_blocks):
        s = b * bs
        for p in range(n_passes):
            out = planes[p]
            for i in range(s, s + bs):
                if src[i] < 0:
                    out[i] = lit[i]
                elif src[i] >= s:
                    out[i] = out[src[i]]
                else:
                    out[i] = pattern(np.uint64(src[i]), p, n_passes)
    final = planes[0].copy()
    for b in range(1, n_blocks):
        s = b * bs
        derived, org = origin([pl[s:s + bs] for pl in planes], n_passes)
        assert (org[derived] < s).all()
        final[s:s + bs][derived] = final[org[derived]]
    assert (final == want).all()
