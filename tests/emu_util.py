"""ctypes access to the CPU SIMT emulator build of the kernels (tests/emu/libtha4_emu.so)."""
import ctypes as C
import os

import numpy as np

from tha4_amd import _capi
from tha4_amd.weights import split_flat_weights

HERE = os.path.dirname(os.path.abspath(__file__))
K_POSEBIAS, K_FACE, K_L0, K_L1, K_L2 = range(5)


class EmuStudent:
    def __init__(self, flat_weights, axes=None, gen=1):
        self.gen = gen
        self.lib = C.CDLL(os.path.join(HERE, "emu", "libtha4_emu.so"))
        L = self.lib
        L.emu_student_create_gen.restype = C.c_void_p
        L.emu_student_create_gen.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.emu_student_grid_gen.argtypes = [C.c_int, C.c_int]
        L.emu_student_block_pixels_gen.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.emu_student_buffer.restype = C.POINTER(C.c_float)
        L.emu_student_buffer.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]
        L.emu_student_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.emu_student_grid.argtypes = [C.c_int]
        L.emu_student_block_pixels.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.emu_student_destroy.argtypes = [C.c_void_p]
        L.emu_sin_omega.restype = C.c_float
        L.emu_sin_omega.argtypes = [C.c_float]
        L.emu_sin_u.restype = C.c_float
        L.emu_sin_u.argtypes = [C.c_float]
        face_sd, body_sd = split_flat_weights(flat_weights)
        ws, self._keep = _capi.build_student_weights(face_sd, body_sd)
        ax, self._keep2 = _capi.build_position_axes(axes)
        self.h = L.emu_student_create_gen(C.byref(ws), C.byref(ax) if ax is not None else None, gen)
        assert self.h

    def buf(self, name):
        n = C.c_int64()
        p = self.lib.emu_student_buffer(self.h, name.encode(), C.byref(n))
        assert p
        return np.ctypeslib.as_array(p, shape=(n.value,))

    def grid(self, kernel):
        return self.lib.emu_student_grid_gen(kernel, self.gen)

    def run(self, kernel, first, count=1):
        assert self.lib.emu_student_run(self.h, kernel, first, count) == 0

    def block_of_tile(self, kernel, tile):
        """Workgroup id that computes the tile-th run of pixels (inverse of the kernels' XCD-aware remap xcd_tile)."""
        n = self.grid(kernel)
        if kernel == 0 or n % 8:
            return tile
        return (tile % (n // 8)) * 8 + tile // (n // 8)

    def block_pixels(self, kernel, block):
        f, c = C.c_int(), C.c_int()
        self.lib.emu_student_block_pixels_gen(kernel, block, self.gen, C.byref(f), C.byref(c))
        return slice(f.value, f.value + c.value)

    def sin_omega(self, z):
        return self.lib.emu_sin_omega(float(z))

    def sin_u(self, u):
        return self.lib.emu_sin_u(float(u))

    @property
    def handoff_scale(self):
        """Generation 2 folds the sine's frequency - omega_0 / 2 pi, the sine takes turns - into the pose-folded biases and the
        z hand-off images."""
        if self.gen != 2:
            return 1.0
        self.lib.emu_sine_scale16.restype = C.c_float
        return float(self.lib.emu_sine_scale16())

    def close(self):
        if self.h:
            self.lib.emu_student_destroy(self.h)
            self.h = None


def pack_z(z, nb):
    """[C, npix] fp64/fp32 -> the kernels' z image [nb][4][npix][4] (flat fp32; siren_layout.h z_offset)."""
    c, npix = z.shape
    zp = np.zeros((nb * 16, npix), np.float32)
    zp[:c] = z
    return zp.reshape(nb, 4, 4, npix).transpose(0, 1, 3, 2).reshape(-1)


def unpack_z(flat, nb, npix, c):
    return flat.reshape(nb, 4, npix, 4).transpose(0, 1, 3, 2).reshape(nb * 16, npix)[:c]
