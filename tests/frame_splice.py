"""Frames spliced from blocks (tests only): the blocks of zstd frames taken apart and put together with Raw / RLE / literals-only
blocks in between -- any sequence of valid blocks is a valid frame, and what it decodes to is for the oracle to say."""


def frame_blocks(frame):
    """the blocks of a zstd frame as [(type, payload bytes, regenerated size for RLE)] (frame.go / block.go)"""
    fhd = frame[4]
    single, dict_flag, fcs_flag = (fhd >> 5) & 1, fhd & 3, fhd >> 6
    pos = 5 + (0 if single else 1) + (0, 1, 2, 4)[dict_flag] + ((1 if single else 0), 2, 4, 8)[fcs_flag]
    blocks = []
    while True:
        h = frame[pos] | (frame[pos + 1] << 8) | (frame[pos + 2] << 16)
        last, typ, size = h & 1, (h >> 1) & 3, h >> 3
        n = 1 if typ == 1 else size
        blocks.append((typ, frame[pos + 3:pos + 3 + n], size))
        pos += 3 + n
        if last:
            return blocks


def splice_frame(blocks):
    """a frame of these blocks: no content size, a window of 128 MiB (window descriptor 0x88)"""
    out = bytearray(b"\x28\xb5\x2f\xfd\x00\x88")
    for i, (typ, payload, size) in enumerate(blocks):
        h = (1 if i + 1 == len(blocks) else 0) | (typ << 1) | (size << 3)
        out += bytes([h & 0xFF, (h >> 8) & 0xFF, h >> 16]) + bytes(payload)
    return bytes(out)


def literal_block(data, rle=False):
    """a compressed block without sequences whose literals are raw (or one repeated byte)"""
    n = len(data)
    lit_type = 1 if rle else 0
    if n < 32:
        lh = bytes([lit_type | (n << 3)])
    elif n < 4096:
        v = lit_type | (1 << 2) | (n << 4)
        lh = bytes([v & 0xFF, v >> 8])
    else:
        v = lit_type | (3 << 2) | (n << 4)
        lh = bytes([v & 0xFF, (v >> 8) & 0xFF, v >> 16])
    body = lh + (data[:1] if rle else data) + b"\x00"
    return (2, body, len(body))


def one_sequence_block(states, literals: bytes, ll: int, ml: int, offset_value: int):
    """A compressed block of Raw literals and ONE sequence under the predefined tables (sequences.go:228-269 modes 0; predefined.go):
    `ll` literals, then a match of `ml` bytes with Offset_Value `offset_value` (1-3: a repeat offset, sequences.go / the offset
    history; above: offset + 3); the literals behind the sequence's `ll` follow the match.  `states`: for LL / OF / ML a map code ->
    a state of the predefined table that decodes to it (see predefined_states).  -> (2, payload, 0) for splice_frame."""
    from tests.desc_interp import LL_BASE, LL_EXTRA, ML_BASE, ML_EXTRA
    llc = max(c for c in range(len(LL_BASE)) if LL_BASE[c] <= ll)
    mlc = max(c for c in range(len(ML_BASE)) if ML_BASE[c] <= ml)
    ofc = offset_value.bit_length() - 1
    assert ll - LL_BASE[llc] < (1 << LL_EXTRA[llc]) and ml - ML_BASE[mlc] < (1 << ML_EXTRA[mlc]) and ofc <= 28
    # what the backward reader reads, in its order (sequences.go:126-226): the three initial states, then the offset's, the match
    # length's and the literal length's extra bits
    reads = [(states[0][llc], 6), (states[1][ofc], 5), (states[2][mlc], 6),
             (offset_value - (1 << ofc), ofc), (ml - ML_BASE[mlc], ML_EXTRA[mlc]), (ll - LL_BASE[llc], LL_EXTRA[llc])]
    acc = nbits = 0
    for v, n in reversed(reads):  # written first = read last
        acc |= v << nbits
        nbits += n
    acc |= 1 << nbits  # the padding marker above the first bits read
    nbits += 1
    stream = acc.to_bytes((nbits + 7) // 8, "little")
    n = len(literals)
    assert n < 4096
    lh = bytes([n << 3]) if n < 32 else bytes([0x04 | ((n & 15) << 4), n >> 4])  # Raw literals: 5-bit / 12-bit size
    payload = lh + bytes(literals) + b"\x01\x00" + stream  # one sequence; all three tables predefined
    return (2, payload, len(payload))


def predefined_states(oracle):
    """for LL, OF, ML: {code: a state of the predefined decoding table (predefined.go:22,52,70) whose cell holds that code}"""
    import ctypes
    from tests import oracle_binding as ob
    out = []
    for kind, log in ((0, 6), (1, 5), (2, 6)):
        t = ob.FseTable()
        assert oracle.lib.orc_fse_build_predefined(ctypes.byref(t), kind) == 0
        m = {}
        for s in range(1 << log):
            m.setdefault(int(t.table[s].raw_symbol), s)
        oracle.lib.orc_fse_free(ctypes.byref(t))
        out.append(m)
    return out
