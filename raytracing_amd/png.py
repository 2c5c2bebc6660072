"""8-bit RGB PNG files, the format the reference writes its result in
(src/main.rs:157-160, image::save_buffer with ColorType::RGB(8)).  Standard
library only: zlib for the IDAT stream and the chunk checksums."""
import struct
import zlib

import numpy as np

_SIGNATURE = b"\x89PNG\r\n\x1a\n"


def _chunk(tag, payload):
    return struct.pack(">I", len(payload)) + tag + payload + struct.pack(">I", zlib.crc32(tag + payload) & 0xFFFFFFFF)


def write_png(path, rgb8):
    """Writes a (ny, nx, 3) uint8 array, top row first, as a non-interlaced
    truecolour PNG (filter type 0 on every row)."""
    a = np.ascontiguousarray(rgb8)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("expected a (ny, nx, 3) uint8 array")
    ny, nx = a.shape[0], a.shape[1]
    rows = np.empty((ny, 1 + nx * 3), dtype=np.uint8)
    rows[:, 0] = 0
    rows[:, 1:] = a.reshape(ny, nx * 3)
    data = _SIGNATURE
    data += _chunk(b"IHDR", struct.pack(">IIBBBBB", nx, ny, 8, 2, 0, 0, 0))
    data += _chunk(b"IDAT", zlib.compress(rows.tobytes(), 6))
    data += _chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(data)


def read_png(path):
    """Reads back what write_png writes (and any other 8-bit, non-interlaced RGB
    or RGBA PNG); returns (ny, nx, 3) uint8."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != _SIGNATURE:
        raise ValueError("not a PNG file")
    at = 8
    idat = []
    nx = ny = channels = None
    while at < len(data):
        (length,) = struct.unpack(">I", data[at:at + 4])
        tag = data[at + 4:at + 8]
        payload = data[at + 8:at + 8 + length]
        (crc,) = struct.unpack(">I", data[at + 8 + length:at + 12 + length])
        if zlib.crc32(tag + payload) & 0xFFFFFFFF != crc:
            raise ValueError("bad chunk checksum")
        if tag == b"IHDR":
            nx, ny, depth, colour, _, _, interlace = struct.unpack(">IIBBBBB", payload)
            if depth != 8 or colour not in (2, 6) or interlace != 0:
                raise ValueError("only 8-bit non-interlaced RGB/RGBA is supported")
            channels = 3 if colour == 2 else 4
        elif tag == b"IDAT":
            idat.append(payload)
        elif tag == b"IEND":
            break
        at += 12 + length
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(ny, 1 + nx * channels)
    out = np.zeros((ny, nx * channels), dtype=np.uint8)
    bpp = channels
    for y in range(ny):
        ftype = raw[y, 0]
        line = raw[y, 1:].astype(np.int32)
        prev = out[y - 1].astype(np.int32) if y else np.zeros(nx * channels, dtype=np.int32)
        if ftype == 0:
            cur = line
        elif ftype == 2:
            cur = (line + prev) & 255
        else:
            cur = np.zeros(nx * channels, dtype=np.int32)
            for i in range(nx * channels):
                left = cur[i - bpp] if i >= bpp else 0
                up = prev[i]
                ul = prev[i - bpp] if i >= bpp else 0
                if ftype == 1:
                    pred = left
                elif ftype == 3:
                    pred = (left + up) >> 1
                elif ftype == 4:
                    p = left + up - ul
                    pa, pb, pc = abs(p - left), abs(p - up), abs(p - ul)
                    pred = left if (pa <= pb and pa <= pc) else (up if pb <= pc else ul)
                else:
                    raise ValueError("bad filter type")
                cur[i] = (line[i] + pred) & 255
        out[y] = cur.astype(np.uint8)
    return out.reshape(ny, nx, channels)[:, :, :3].copy()
