"""CPU check of the arithmetic the HIP kernels execute: csrc/gvp_math.h is
compiled with g++ into a throw-away test harness (tests/host_math/host_lba.cpp)
and compared with the reference's golden vectors.  The harness is scaffolding,
not a product path: caster-dta_amd never loads it."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from conftest import REPO, rel_err
from gvp_hip import arena

SRC = os.path.join(REPO, "tests", "host_math", "host_lba.cpp")


@pytest.fixture(scope="module")
def host_lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("host_math") / "host_lba.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", so, SRC])
    lib = C.CDLL(so)
    lib.host_lba_forward.restype = C.c_int
    return lib


def _run(lib, flat, g, num_convs=2, mean=0, nt=(20, 1)):
    N, E = g["x_s"].shape[0], g["e_s"].shape[0]
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    i = lambda a: np.ascontiguousarray(a, dtype=np.int64)
    arrs = dict(P=f(flat), x_s=f(g["x_s"]), x_v=f(g["x_v"]), nt=i(g["ntypes"]), e_s=f(g["e_s"]),
                e_v=f(g["e_v"]), et=i(g["etypes"]), ei=i(g["edge_index"]))
    out = np.zeros((N, 64), np.float32)
    sh = np.zeros((num_convs + 1, N, 28), np.float32)
    sdh = np.zeros((num_convs, N, 28), np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib.host_lba_forward(p(arrs["P"]), nt[0], nt[1], num_convs, mean, p(arrs["x_s"]), p(arrs["x_v"]),
                              p(arrs["nt"]), p(arrs["e_s"]), p(arrs["e_v"]), p(arrs["et"]), p(arrs["ei"]),
                              C.c_int64(N), C.c_int64(E), p(out), p(sh), p(sdh))
    assert rc == flat.size, rc
    return out, sh, sdh


def test_kernel_math_matches_reference(host_lib, lba_small, protein_params):
    g = lba_small
    flat = arena.flatten_state(protein_params, 2).numpy()
    assert flat.size == 15117
    out, sh, sdh = _run(host_lib, flat, g)
    tol = 1e-5
    N = g["x_s"].shape[0]
    merged = lambda name: np.concatenate([g[f"stage_{name}_s"], g[f"stage_{name}_v"].reshape(N, -1)], 1)
    assert rel_err(sh[0], merged("node_embed")) < tol
    assert rel_err(sdh[0], merged("conv0_dh")) < tol
    assert rel_err(sh[1], merged("conv0")) < tol
    assert rel_err(sdh[1], merged("conv1_dh")) < tol
    assert rel_err(sh[2], merged("conv1")) < tol
    assert rel_err(out, g["out"]) < tol
    assert rel_err(out, g["out64"]) < tol


# ------------------------------------------------------------------ in-kernel dropout generator (csrc/gvp_rng.h)
@pytest.fixture(scope="module")
def host_rng(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("host_rng") / "host_rng.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", so,
                           os.path.join(REPO, "tests", "host_math", "host_rng.cpp")])
    return C.CDLL(so)


def test_philox_known_answers(host_rng):
    """Philox4x32-10 known-answer vectors published with Random123 (kat_vectors: all-zero, all-ones, pi digits)."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        c, k, o = (C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), (C.c_uint32 * 4)()
        host_rng.host_philox(c, k, o)
        assert tuple(o) == want


def test_dropout_factors_host(host_rng):
    N, W, p = 4000, 20, 0.2
    out = np.zeros((N, W), np.float32)
    host_rng.host_dropout_mask(C.c_ulonglong(1234567), C.c_ulonglong(99), 3, C.c_longlong(N), W, C.c_float(p),
                               out.ctypes.data_as(C.c_void_p))
    assert set(np.unique(out).round(4)) == {0.0, 1.25}
    assert abs((out == 0).mean() - p) < 0.01
    again = np.zeros_like(out)
    host_rng.host_dropout_mask(C.c_ulonglong(1234567), C.c_ulonglong(99), 3, C.c_longlong(N), W, C.c_float(p),
                               again.ctypes.data_as(C.c_void_p))
    assert np.array_equal(out, again)
    other = np.zeros_like(out)
    host_rng.host_dropout_mask(C.c_ulonglong(1234567), C.c_ulonglong(99), 4, C.c_longlong(N), W, C.c_float(p),
                               other.ctypes.data_as(C.c_void_p))
    assert (out != other).mean() > 0.2
