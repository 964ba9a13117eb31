"""TEST INFRASTRUCTURE: drive the emulator build of the C ABI (tests/emu/_build/libdvs_emu.so) with numpy buffers."""
import ctypes

import numpy as np

from dags_vae_search_amd import _lib as dl
from tests.emu import build as emu_build

_emu = None


def emu():
    global _emu
    if _emu is None:
        _emu = dl.bind(ctypes.CDLL(emu_build.build()))
    return _emu


def ptr(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


class EmuModel:
    """Flat parameter buffer + workspace for one (N, C, batch) problem on the emulator."""

    def __init__(self, cfg, params, batch, training=False, dropout=0.15, seed=0, dag_offset=0, beta=0.005):
        self.lib = emu()
        self.cfg = cfg
        self.shape = dl.make_shape(batch, cfg.N, cfg.C, training, dropout, beta, 0.01, dag_offset, seed)
        self.table, self.P = dl.param_table(self.lib, self.shape)
        self.flat = np.zeros(self.P, np.float32)
        for name, off, shp in self.table:
            v = np.asarray(params[name], np.float32).reshape(-1)
            self.flat[off:off + v.size] = v
        self.ws = np.zeros(self.lib.dvs_workspace_bytes(ctypes.byref(self.shape)) // 4 + 64, np.float32)
        self.record_bytes = dl.record_bytes(self.lib, self.shape)
        self.records = np.zeros(batch * self.record_bytes, np.uint8)
        self.batch = batch

    def pack(self, feats):
        lab = np.ascontiguousarray(feats["vertex_label_features"], np.float32)
        pos = np.ascontiguousarray(feats["vertex_position_features"], np.float32)
        adj = np.ascontiguousarray(feats["adjacency_matrices"], np.float32)
        tm = np.ascontiguousarray(feats["target_masks"]).astype(np.uint8)
        status = np.zeros(1, np.int32)
        dl.check(self.lib, self.lib.dvs_pack_features(ctypes.byref(self.shape), ptr(lab), ptr(pos), ptr(adj), ptr(tm),
                                                      ptr(self.records), self.records.nbytes, ptr(status), None), "pack")
        self.status = status
        return int(status[0])

    def forward(self, eps=None):
        losses = np.zeros(dl.LOSS_FLOATS, np.float32)
        mu = np.zeros((self.batch, 32), np.float32)
        lv = np.zeros((self.batch, 32), np.float32)
        e = None if eps is None else np.ascontiguousarray(eps, np.float32)
        dl.check(self.lib, self.lib.dvs_loss_forward(ctypes.byref(self.shape), ptr(self.records), self.records.nbytes,
                                                     ptr(self.flat), self.flat.size, ptr(self.ws), self.ws.nbytes, ptr(e),
                                                     ptr(getattr(self, "status", None)), ptr(losses), ptr(mu), ptr(lv),
                                                     None), "forward")
        return losses, mu, lv

    def backward(self, g_recon=1.0, g_kld=0.005, clip_scratch=None):
        """clip_scratch (float32[CLIP_SCRATCH_FLOATS]): dvs_loss_backward_sq — the slab reduction also leaves the gradient's
        partial sums of squares in clip_scratch[2:] for dvs_clip_adam_from_partials."""
        gcoef = np.asarray([g_recon, g_kld], np.float32)
        grads = np.full(self.P, np.nan, np.float32)
        if clip_scratch is not None:
            dl.check(self.lib, self.lib.dvs_loss_backward_sq(ctypes.byref(self.shape), ptr(self.records), self.records.nbytes,
                                                             ptr(self.flat), self.flat.size, ptr(self.ws), self.ws.nbytes,
                                                             ptr(gcoef), ptr(grads), ptr(clip_scratch), None), "backward_sq")
        else:
            dl.check(self.lib, self.lib.dvs_loss_backward(ctypes.byref(self.shape), ptr(self.records), self.records.nbytes,
                                                          ptr(self.flat), self.flat.size, ptr(self.ws), self.ws.nbytes,
                                                          ptr(gcoef), ptr(grads), None), "backward")
        return {name: grads[off:off + int(np.prod(shp))].reshape(shp) for name, off, shp in self.table}, grads

    def activation(self, slot):
        out = np.zeros((self.batch, 16 * ((self.cfg.N + 15) // 16), 64), np.float32)
        dl.check(self.lib, self.lib.dvs_debug_activation(ctypes.byref(self.shape), ptr(self.ws), slot, ptr(out), None),
                 "debug_activation")
        return out
