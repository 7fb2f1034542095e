"""Minimal PNG codec for the trajectory layout (8-bit RGB, no interlace).

The reference writes ``images{c}/im_{t}.png`` with OpenCV (``visual_mpc/sim/simulator.py:82-86``,
``cv2.imwrite(..., images[t, i, :, :, ::-1])`` - the ``::-1`` undoes OpenCV's BGR convention, so the
file holds the frame in RGB).  OpenCV is not part of this stack; a PNG of this kind is a zlib
stream of filter-0 scanlines between three chunks, which the standard library covers.
"""
import struct
import zlib

import numpy as np

_SIG = b'\x89PNG\r\n\x1a\n'


def _chunk(tag, data):
    return struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(tag + data) & 0xffffffff)


def write_png(path, rgb):
    """Write a uint8 ``[H, W, 3]`` RGB array."""
    rgb = np.ascontiguousarray(rgb)
    if rgb.dtype != np.uint8 or rgb.ndim != 3 or rgb.shape[2] != 3:
        raise ValueError('write_png takes a uint8 [H, W, 3] array, got %s %s' % (rgb.dtype, rgb.shape))
    h, w = rgb.shape[:2]
    raw = np.zeros((h, 1 + 3 * w), np.uint8)        # filter byte 0 ("None") in front of every scanline
    raw[:, 1:] = rgb.reshape(h, 3 * w)
    with open(path, 'wb') as f:
        f.write(_SIG)
        f.write(_chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 2, 0, 0, 0)))
        f.write(_chunk(b'IDAT', zlib.compress(raw.tobytes(), 6)))
        f.write(_chunk(b'IEND', b''))


def read_png(path):
    """Read back a PNG written by ``write_png`` (8-bit RGB, filter 0 only) -> uint8 ``[H, W, 3]``."""
    with open(path, 'rb') as f:
        blob = f.read()
    if blob[:8] != _SIG:
        raise ValueError('%s is not a PNG file' % path)
    pos, idat, shape = 8, b'', None
    while pos < len(blob):
        n, tag = struct.unpack('>I4s', blob[pos:pos + 8])
        data = blob[pos + 8:pos + 8 + n]
        if struct.unpack('>I', blob[pos + 8 + n:pos + 12 + n])[0] != (zlib.crc32(tag + data) & 0xffffffff):
            raise ValueError('bad CRC in chunk %r' % tag)
        if tag == b'IHDR':
            w, h, depth, ctype, _, _, interlace = struct.unpack('>IIBBBBB', data)
            if (depth, ctype, interlace) != (8, 2, 0):
                raise ValueError('only 8-bit non-interlaced RGB is supported')
            shape = (h, w)
        elif tag == b'IDAT':
            idat += data
        pos += 12 + n
    h, w = shape
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    if raw[:, 0].any():
        raise ValueError('only filter type 0 scanlines are supported')
    return raw[:, 1:].reshape(h, w, 3).copy()
