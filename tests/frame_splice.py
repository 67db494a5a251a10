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
