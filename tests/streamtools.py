"""Parse packer streams for test diagnostics."""
import struct


def parse_stream(s, hdr_len=0):
    """-> dict(method, header, planes=[dict(len, in_size, blocks=[(mode, payload_len, crc, offset)])])"""
    out = {"method": s[0], "header": bytes(s[1 : 1 + hdr_len]), "planes": []}
    pos = 1 + hdr_len
    while pos < len(s):
        (ln,) = struct.unpack_from("<I", s, pos)
        (n,) = struct.unpack_from("<I", s, pos + 4)
        blocks, q, left = [], pos + 8, n
        while left > 0:
            plen = struct.unpack_from("<H", s, q)[0] + 1
            crc = struct.unpack_from("<I", s, q + 2)[0]
            blocks.append((s[q + 6], plen, crc, q))
            q += 7 + plen
            left -= min(left, 65536)
        out["planes"].append({"len": ln, "in_size": n, "blocks": blocks, "offset": pos})
        pos += 4 + ln
    out["end"] = pos
    return out


def describe_mismatch(got, want, hdr_len=0):
    if got == want:
        return "identical"
    msg = ["len got %d want %d" % (len(got), len(want))]
    n = min(len(got), len(want))
    first = next((i for i in range(n) if got[i] != want[i]), n)
    msg.append("first differing byte at %d" % first)
    try:
        pw = parse_stream(want, hdr_len)
        for k, pl in enumerate(pw["planes"]):
            for j, (mode, plen, crc, off) in enumerate(pl["blocks"]):
                if off <= first < off + 7 + plen:
                    msg.append("inside plane %d block %d (mode %d, payload %d, block offset %d, +%d)" % (k, j, mode, plen, off, first - off))
        pg = parse_stream(got, hdr_len)
        msg.append("want planes %s" % [(p["len"], [(b[0], b[1]) for b in p["blocks"]][:4]) for p in pw["planes"]])
        msg.append("got  planes %s" % [(p["len"], [(b[0], b[1]) for b in p["blocks"]][:4]) for p in pg["planes"]])
    except Exception as e:  # a corrupt stream must not hide the original failure
        msg.append("(parse failed: %r)" % (e,))
    msg.append("want[%d:%d]=%s" % (max(0, first - 4), first + 12, want[max(0, first - 4) : first + 12].hex()))
    msg.append("got [%d:%d]=%s" % (max(0, first - 4), first + 12, got[max(0, first - 4) : first + 12].hex()))
    return "; ".join(msg)
