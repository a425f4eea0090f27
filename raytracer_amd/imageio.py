"""Image output and progressive-render checkpoints.

The reference writes one 8-bit RGB PNG at the end of main() (src/main.rs:226,240: `img.put_pixel(x, y, ...)`,
`img.save("output/test.png")`, rows in loop order, row 0 first) and keeps nothing in between. Here the float
accumulation buffer can also be saved with the number of samples it holds, so that a long render can stop and
resume: samples are keyed by their index, so samples [0, a) + [a, b) give the same sums as [0, b) when the batch
boundaries are the same."""
import json
import os
import struct
import zlib

import numpy as np


def _chunk(tag, data):
    body = tag + data
    return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)


def encode_png(rgb):
    """8-bit RGB PNG (colour type 2, no interlace, filter 0 on every row) of a uint8 [H, W, 3] array."""
    rgb = np.ascontiguousarray(rgb)
    if rgb.dtype != np.uint8 or rgb.ndim != 3 or rgb.shape[2] != 3:
        raise ValueError("encode_png wants uint8 [H, W, 3]")
    h, w, _ = rgb.shape
    raw = np.empty((h, 1 + 3 * w), dtype=np.uint8)
    raw[:, 0] = 0
    raw[:, 1:] = rgb.reshape(h, 3 * w)
    return (b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
            _chunk(b"IDAT", zlib.compress(raw.tobytes(), 6)) + _chunk(b"IEND", b""))


def decode_png(data):
    """Inverse of encode_png for files it wrote (8-bit RGB, filter 0); used by the tests."""
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG")
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        (n,) = struct.unpack(">I", data[pos:pos + 4])
        tag = data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        if zlib.crc32(tag + body) & 0xFFFFFFFF != crc:
            raise ValueError("bad CRC in chunk %r" % tag)
        if tag == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
            if (depth, ctype, interlace) != (8, 2, 0):
                raise ValueError("only 8-bit RGB, non-interlaced")
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + 3 * w)
    if raw[:, 0].any():
        raise ValueError("only filter type 0")
    return raw[:, 1:].reshape(h, w, 3).copy()


def encode_ppm(rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    return b"P6\n%d %d\n255\n" % (w, h) + rgb.tobytes()


def write_image(path, rgb):
    """PNG, or binary PPM when the name ends in .ppm."""
    data = encode_ppm(rgb) if path.lower().endswith(".ppm") else encode_png(rgb)
    with open(path, "wb") as f:
        f.write(data)


def save_checkpoint(path, accum, spp_done, meta):
    """accum: float32 [H, W, 3] sum over the first spp_done samples of every pixel; meta: what the sums depend on
    (scene, seed, size, batch size ...), compared on resume."""
    # written beside the target and moved into place: a run that is interrupted while saving leaves the previous
    # checkpoint intact instead of a truncated file
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        np.savez(f, accum=np.asarray(accum, dtype=np.float32), spp_done=np.int64(spp_done), meta=np.frombuffer(json.dumps(meta, sort_keys=True).encode(), dtype=np.uint8))
        f.flush()
        os.fsync(f.fileno())
    os.replace(tmp, path)


def load_checkpoint(path, meta):
    """Returns (accum, spp_done). Raises if the checkpoint was made with other parameters."""
    with np.load(path, allow_pickle=False) as z:
        saved = json.loads(bytes(z["meta"]).decode())
        if saved != json.loads(json.dumps(meta, sort_keys=True)):
            diff = sorted(k for k in set(saved) | set(meta) if saved.get(k) != meta.get(k))
            raise ValueError("checkpoint %s was made with different parameters: %s" % (path, ", ".join(diff)))
        return z["accum"].astype(np.float32), int(z["spp_done"])
