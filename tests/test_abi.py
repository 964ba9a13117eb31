"""The C-ABI library loads on a CPU-only box and exports every symbol include/dvs.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from dags_vae_search_amd import _lib as dl

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(REPO, "include", "dvs.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dvs_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(dl.lib_path()):
        pytest.fail(f"{dl.lib_path()} is missing: run __graft_entry__.build()")
    lib = ctypes.CDLL(dl.lib_path())
    names = header_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(dl.EXPORTS)          # the Python binding covers the whole header


def test_host_side_queries_without_a_gpu():
    lib = dl.load()
    assert lib.dvs_version() == 202 == dl.ABI_VERSION
    shape = dl.make_shape(4096, 15, 15)
    table, total = dl.param_table(lib, shape)
    assert len(table) == 108 and total % 4 == 0
    assert table[0][0] == "vertex_position_embed.W1" and table[-1][0] == "fc3.bias"
    assert all(off % 4 == 0 for _, off, _ in table)
    assert sum(int(__import__("numpy").prod(s)) for _, _, s in table) == 310160       # n=12 card=12
    assert lib.dvs_workspace_bytes(ctypes.byref(shape)) > 0
    assert dl.record_bytes(lib, shape) == 96
    alarm = dl.make_shape(16, 40, 40)                                                   # alarm size: wide path
    assert lib.dvs_param_count(ctypes.byref(alarm)) >= 470185                           # n=37 card=37
    table40, _ = dl.param_table(lib, alarm)
    assert sum(int(__import__("numpy").prod(s)) for _, _, s in table40) == 470185
    assert dl.record_bytes(lib, alarm) == 864
    bad = dl.make_shape(16, 49, 40)
    assert lib.dvs_param_count(ctypes.byref(bad)) < 0
    assert b"n_tokens" in lib.dvs_last_error()


def test_model_refuses_cpu_compute_and_keeps_state_dict_contract():
    import numpy as np
    import torch
    from dags_vae_search_amd import PaceVaeV3
    torch.manual_seed(9)
    m = PaceVaeV3(12, 12, 32, 8, 3, 64, 32, 32, 0.15)
    z = np.load(os.path.join(REPO, "tests", "golden", "golden_n12c12.npz"))
    sd = m.state_dict()
    # same module tree and init order as the reference => identical tensors under the same torch seed
    for k, v in sd.items():
        assert np.array_equal(v.numpy(), z["param/" + k]), k
    assert m.max_num_vertices == 15 and m.vertex_label_cardinality == 15
    with pytest.raises(RuntimeError):
        m.encode_direct({})
    # parameters alias one flat buffer
    p = next(m.parameters())
    assert p.data_ptr() == m.flat_params.data_ptr()


def test_undersized_caller_buffers_are_rejected_before_anything_is_enqueued():
    """include/dvs.h, code 14: records_bytes / n_params / workspace_bytes / state_bytes below what the shape needs come
    back as an error (with the needed size in dvs_last_error) instead of a silent out-of-bounds device access.  The check
    runs before any HIP call, so it is testable on a CPU-only box with dummy non-null pointers."""
    lib = dl.load()
    shape = dl.make_shape(64, 15, 15)
    sb = ctypes.byref(shape)
    rec_need = 64 * dl.record_bytes(lib, shape)
    ws_need = lib.dvs_workspace_bytes(sb)
    n_params = lib.dvs_param_count(sb)
    dummy = ctypes.c_void_p(4096)           # never dereferenced: every call below fails validation first

    def last():
        return lib.dvs_last_error().decode()
    assert lib.dvs_pack_features(sb, dummy, dummy, dummy, dummy, dummy, rec_need - 1, dummy, None) == 14
    assert "records_bytes" in last() and str(rec_need) in last()
    assert lib.dvs_build_records(sb, dummy, dummy, dummy, rec_need - 1, dummy, None) == 14
    fwd = lambda rb, npar, wb: lib.dvs_loss_forward(sb, dummy, rb, dummy, npar, dummy, wb, None, None, dummy, None, None,
                                                    None)
    assert fwd(rec_need - 1, n_params, ws_need) == 14 and "records_bytes" in last()
    assert fwd(rec_need, n_params - 1, ws_need) == 14 and "n_params" in last()
    assert fwd(rec_need, n_params, ws_need - 4) == 14 and "workspace_bytes" in last() and str(ws_need) in last()
    assert lib.dvs_loss_backward(sb, dummy, rec_need, dummy, n_params, dummy, ws_need - 4, dummy, dummy, None) == 14
    assert lib.dvs_loss_backward(sb, dummy, rec_need, dummy, n_params - 1, dummy, ws_need, dummy, dummy, None) == 14
    assert lib.dvs_encode(sb, dummy, rec_need, dummy, n_params, dummy, ws_need - 4, dummy, dummy, None) == 14
    assert lib.dvs_decode(sb, dummy, n_params, dummy, ws_need, dummy, rec_need, dummy, None, dummy,
                          64 * dl.DECODE_STATE_BYTES - 1, None) == 14
    assert "state_bytes" in last()
    assert lib.dvs_decode(sb, dummy, n_params, dummy, ws_need - 4, dummy, rec_need, dummy, None, dummy,
                          64 * dl.DECODE_STATE_BYTES, None) == 14
    # a bigger batch against buffers sized for a smaller one: the classic mistake this argument exists for
    big = dl.make_shape(128, 15, 15)
    assert lib.dvs_loss_forward(ctypes.byref(big), dummy, rec_need, dummy, n_params, dummy, ws_need, None, None, dummy, None,
                                None, None) == 14
    # null pointers are still code 10 and bad shapes 1..5
    assert lib.dvs_loss_forward(sb, None, rec_need, dummy, n_params, dummy, ws_need, None, None, dummy, None, None, None) == 10
