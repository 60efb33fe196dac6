"""ctypes binding of libcaster_gvp.so (C ABI: include/caster_gvp.h).

The library is the product: there is no Python / CPU fallback.  Importing this
module never fails (so CPU-only tooling can introspect the package), but the
first call that needs a kernel raises `HipLibraryError` when the shared object
is missing or stale.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must be imported first: its libamdhip64.so.7 is the HIP runtime we bind to)

_HERE = os.path.dirname(os.path.abspath(__file__))
# CGVP_LIB_PATH: A/B builds of the same ABI (diagnostics); the default is the in-tree library
LIB_PATH = os.environ.get("CGVP_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "lib", "libcaster_gvp.so")
ABI_VERSION = 29
# sha256 of include/caster_gvp.h with comments, blank space and the CGVP_ABI_VERSION line removed, as
# `abi_header_digest()` computes it.  tests/test_abi.py fails when the header's declarations change
# without CGVP_ABI_VERSION, ABI_VERSION and this digest being updated together.
ABI_HEADER_SHA256 = "362eacc64a468f7a3f91503a8f667d128e3a4dbc6a5b29f605a976517cde6195"


class HipLibraryError(RuntimeError):
    pass


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "node_in_s", "node_in_v", "edge_in_s", "edge_in_v", "hidden_s", "hidden_v",
        "edge_hidden_s", "edge_hidden_v", "out_s", "storage", "layer_kind")]


class Layout(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "nt_node", "nt_edge", "node_gvp", "node_ln", "edge_gvp", "edge_ln", "conv0",
        "conv_stride", "ln_out", "head", "total")]


class Segment(C.Structure):
    _fields_ = [("slab", C.c_void_p), ("rows", C.c_int32), ("stride", C.c_int32), ("col0", C.c_int32),
                ("len", C.c_int32), ("dst", C.c_int32)]


class Rng(C.Structure):
    """cgvp_rng: in-kernel dropout -- device pointer to {seed, offset} (2 x uint64), drop probability, first stream id."""
    _fields_ = [("seed", C.c_void_p), ("p", C.c_float), ("stream", C.c_int32)]


class AttnProblem(C.Structure):
    """cgvp_attn_problem (one direction of the varlen cross attention)."""
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("q_ptr", C.c_void_p), ("k_ptr", C.c_void_p),
                ("num_q", C.c_int64), ("num_k", C.c_int64), ("out", C.c_void_p), ("lse", C.c_void_p),
                ("g_out", C.c_void_p), ("delta", C.c_void_p), ("g_q", C.c_void_p), ("g_k", C.c_void_p),
                ("g_v", C.c_void_p), ("weights", C.c_void_p), ("weights_lq", C.c_int64), ("weights_lk", C.c_int64)]


class GineW(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("eps", "we", "be", "w0", "b0", "w1", "b1")]


MAX_STAGE = 24           # CGVP_MAX_STAGE


class StageItem(C.Structure):
    """cgvp_stage_item: one copy-and-pad of cgvp_stage_buffers."""
    _fields_ = [("dst", C.c_void_p), ("src", C.c_void_p), ("copy_bytes", C.c_int64), ("capacity_bytes", C.c_int64),
                ("fill_word", C.c_uint32)]


class LbaBatch(C.Structure):
    """cgvp_lba_batch: one batched protein graph as the whole-pass entry points take it."""
    _fields_ = [("num_nodes", C.c_int64), ("num_edges", C.c_int64)] + [(n, C.c_void_p) for n in (
        "x_s", "x_v", "ntypes", "e_s", "e_v", "etypes", "edge_index", "rowptr", "eperm", "esrc", "edst")]


class LbaFwdWs(C.Structure):
    """cgvp_lba_fwd_ws: byte offsets of the sub-buffers of a protein forward workspace."""
    _fields_ = [(n, C.c_int64) for n in ("seed", "image", "state", "e_emb", "rowptr", "eperm", "esrc", "edst",
                                         "ids_scratch", "total", "state_rows", "node_stride")]


GINE_MAX_LAYERS = 8      # CGVP_GINE_MAX_LAYERS


class GineCfg(C.Structure):
    _fields_ = [("num_layers", C.c_int32), ("widths", C.c_int32 * (GINE_MAX_LAYERS + 1)), ("num_ntypes", C.c_int32),
                ("num_etypes", C.c_int32), ("edge_dim", C.c_int32), ("act_slope", C.c_float)]


class GineBatch(C.Structure):
    _fields_ = [("num_nodes", C.c_int64), ("num_edges", C.c_int64)] + [(n, C.c_void_p) for n in (
        "x", "ntypes", "eattr", "etypes", "edge_index", "rowptr", "eperm", "esrc", "edst")]


class GineFwdWs(C.Structure):
    _fields_ = [("seed", C.c_int64), ("hidden", C.c_int64 * GINE_MAX_LAYERS), ("agg", C.c_int64 * GINE_MAX_LAYERS),
                ("pos", C.c_int64 * GINE_MAX_LAYERS)] + [(n, C.c_int64) for n in (
        "rowptr", "eperm", "esrc", "edst", "saved", "ids_scratch", "total")]


_P = C.c_void_p
_I64 = C.c_int64
_I32 = C.c_int32

_SIGNATURES = {
    "cgvp_abi_version": (C.c_int, []),
    "cgvp_build_info": (C.c_char_p, []),
    "cgvp_csr_from_coo": (C.c_int, [_P, _I64, _I64, _P, _P, _P, _P, _P, _I32, _P, _P]),
    "cgvp_csr_collate": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _P, _P, _P, _P, _P]),
    "cgvp_lba_layout": (C.c_int, [C.POINTER(Dims), _I32, _I32, _I32, C.POINTER(Layout)]),
    "cgvp_lba_image_floats": (C.c_int64, [C.POINTER(Dims), C.POINTER(Layout)]),
    "cgvp_lba_prepare": (C.c_int, [C.POINTER(Dims), C.POINTER(Layout), _P, _P, _P]),
    "cgvp_lba_pass_begin": (C.c_int, [C.POINTER(Dims), C.POINTER(Layout), _P, _P, _P, _P, _P, _I64, _P, _P, _P, _P, _I64, _P, _P]),
    "cgvp_node_embed_fwd": (C.c_int, [C.POINTER(Dims), C.POINTER(Layout), _P, _P, _P, _P, _P, _I64, _P, _P, _P, _P]),
    "cgvp_rng_next": (C.c_int, [_P, _P, _P]),
    "cgvp_conv_fwd": (C.c_int, [C.POINTER(Dims), C.POINTER(Layout), _P, _P, _I32, _P, _P, _P, _P, _P, _P, _P,
                                _P, _I64, _I64, _I32, _P, _P, _P, _P]),
    "cgvp_node_update_fwd": (C.c_int, [C.POINTER(Dims), C.POINTER(Layout), _P, _P, _I32, _P, _P, _I64, _I32,
                                       _P, _P, _P]),
    "cgvp_dropout_masks": (C.c_int, [C.POINTER(Rng), _I32, _I64, _I32, _P, _P]),
    "cgvp_conv_layer_fwd": (C.c_int, [C.POINTER(Dims), C.POINTER(Layout), _P, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _I64,
                                      _I64, _I32, _P, _P, C.POINTER(Rng), _I32, _P, _P, _P, _P, _P, _P]),
    "cgvp_node_update_fwd_train": (C.c_int, [C.POINTER(Dims), C.POINTER(Layout), _P, _I32, _P, _P, _P, _P,
                                             C.POINTER(Rng), _I64, _I32, _P, _P, _P]),
    "cgvp_bwd_workspace_floats": (C.c_int64, [C.POINTER(Dims), C.POINTER(Layout)]),
    "cgvp_node_update_bwd": (C.c_int, [C.POINTER(Dims), C.POINTER(Layout), _P, _I32, _P, _P, _P, _P, C.POINTER(Rng), _P, _P,
                                       _P, _P, _P, _I64, _I32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "cgvp_conv_bwd": (C.c_int, [C.POINTER(Dims), C.POINTER(Layout), _P, _I32, _P, _P, _P, _P, _P, _I64,
                                _I64, _I32, _P, _P, _I32, _P, _P, _P, _P, _P, _P, _P]),
    "cgvp_edge_embed_bwd": (C.c_int, [C.POINTER(Dims), C.POINTER(Layout), _P, _P, _P, _P, _P, _I64, C.POINTER(_P), _I32,
                                      _P, _P, _P, _P, _P, _P, _P]),
    "cgvp_node_embed_bwd": (C.c_int, [C.POINTER(Dims), C.POINTER(Layout), _P, _P, _P, _P, _I64, _P, _P, _P, _P, _P,
                                      _P, _P, _P, _P, _P]),
    "cgvp_bwd_reduce": (C.c_int, [_P, _I32, _P, _I32, _P]),
    "cgvp_gine_conv_fwd": (C.c_int, [_P, _P, _I32, _P, _P, _I32, _I32, _P, _P, _P, _P, _I64, _I64, _I32, _I32,
                                     _I32, C.POINTER(GineW), C.c_float, _P, C.POINTER(Rng), _I32, _P, _P]),
    "cgvp_edge_featurise": (C.c_int, [_P, _P, _P, _I64, _I64, _P, _P, _P]),
    "cgvp_attn_fwd": (C.c_int, [C.POINTER(AttnProblem), _I32, _I64, _I32, C.c_float, _P]),
    "cgvp_attn_bwd": (C.c_int, [C.POINTER(AttnProblem), _I32, _I64, _I32, C.c_float, _P]),
    "cgvp_attn_weights": (C.c_int, [C.POINTER(AttnProblem), _I32, _I64, _I32, C.c_float, _P]),
    "cgvp_lba_fwd_workspace": (C.c_int, [C.POINTER(Dims), C.POINTER(Layout), _I64, _I64, _I32, C.POINTER(LbaFwdWs)]),
    "cgvp_lba_forward_pass": (C.c_int, [C.POINTER(Dims), C.POINTER(Layout), _P, C.POINTER(LbaBatch), _I32, C.c_float, _P, _P,
                                        _P, _P, _I32, _I32, _P, _P]),
    "cgvp_lba_forward_plan": (C.c_int, [_I64, _I64, _I32, _I32, _I32, C.POINTER(_I32)]),
    "cgvp_lba_bwd_workspace_bytes": (C.c_int64, [C.POINTER(Dims), C.POINTER(Layout), _I64, _I64]),
    "cgvp_lba_backward_pass": (C.c_int, [C.POINTER(Dims), C.POINTER(Layout), C.POINTER(LbaBatch), _I32, C.c_float, _P, _P,
                                         _P, _P, _P, _P, _P, _P, _P, _P]),
    "cgvp_gine_fwd_workspace": (C.c_int, [C.POINTER(GineCfg), _I64, _I64, _I32, C.POINTER(GineFwdWs)]),
    "cgvp_gine_forward_pass": (C.c_int, [C.POINTER(GineCfg), C.POINTER(GineW), C.POINTER(GineBatch), C.c_float, _P, _P, _P,
                                         _P, _I32, _I32, _P, _P]),
    "cgvp_gine_bwd_workspace_bytes": (C.c_int64, [C.POINTER(GineCfg), _I64, _I64]),
    "cgvp_gine_backward_pass": (C.c_int, [C.POINTER(GineCfg), C.POINTER(GineW), C.POINTER(GineBatch), C.c_float, _P, _P,
                                          _P, _P, _P, _P, _I32, _P]),
    "cgvp_linear_wgrad_workspace_floats": (C.c_int64, [_I64, _I32, _I32]),
    "cgvp_linear_wgrad": (C.c_int, [_P, _P, _I64, _I32, _I32, _P, _P, _P]),
    "cgvp_layer_norm_fwd": (C.c_int, [_P, _P, _P, _I64, _I32, C.c_float, _P, _P, _P, _P]),
    "cgvp_layer_norm_bwd_workspace_floats": (C.c_int64, [_I64, _I32]),
    "cgvp_layer_norm_bwd": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I32, _P, _P, _P, _P]),
    "cgvp_dropout_add": (C.c_int, [_P, _P, C.POINTER(Rng), _I64, _I32, _P, _P]),
    "cgvp_dropout_scale": (C.c_int, [_P, C.POINTER(Rng), _I64, _I32, _P, _P]),
    "cgvp_act_dropout_fwd": (C.c_int, [_P, C.POINTER(Rng), C.c_float, _I64, _I32, _P, _P]),
    "cgvp_act_dropout_bwd": (C.c_int, [_P, _P, C.POINTER(Rng), C.c_float, _I64, _I32, _P, _P]),
    "cgvp_stage_buffers": (C.c_int, [C.POINTER(StageItem), _I32, _P]),
    "cgvp_debug_kernel_timing": (C.c_int, [_I32]),
    "cgvp_debug_kernel_times": (C.c_int, [_P, _P, _I32]),
    "cgvp_gine_bwd_workspace_floats": (C.c_int64, []),
    "cgvp_gine_conv_bwd": (C.c_int, [_P, _P, _I32, _P, _P, _I32, _I32, _P, _P, _P, _P, _I64, _I64, _I32, _I32, _I32,
                                     C.POINTER(GineW), C.c_float, _P, C.POINTER(Rng), _P, _P, _P, _P, _I32, _P]),
}

_lib = None


def abi_header_digest(path=None):
    """Digest of the declarations in include/caster_gvp.h (comments / whitespace / the version define do not
    count)."""
    import hashlib
    import re
    path = path or os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "caster_gvp.h")
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    text = "\n".join(l for l in text.splitlines() if "CGVP_ABI_VERSION" not in l)
    return hashlib.sha256(re.sub(r"\s+", "", text).encode()).hexdigest()


def exported_symbols():
    """Names include/caster_gvp.h declares (the CPU test-suite checks the .so exports all)."""
    return tuple(_SIGNATURES)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  caster-dta_amd has no CPU or PyTorch fallback.")
        try:
            handle = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise HipLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in _SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise HipLibraryError(f"{LIB_PATH} does not export {name}; rebuild the library") from e
            fn.restype, fn.argtypes = res, args
        if handle.cgvp_abi_version() != ABI_VERSION:
            raise HipLibraryError(f"{LIB_PATH} has ABI {handle.cgvp_abi_version()}, binding expects {ABI_VERSION}")
        _lib = handle
    return _lib


_bridge = False


def bridge():
    """The C++ eager fast path (csrc/torch_bridge.cpp -> lib/caster_gvp_torch.so): C++ autograd nodes over the whole-pass
    entry points.  None when it is not built or CGVP_BRIDGE=0 -- the torch.library custom ops (the same kernels, driven
    from Python) are then used in eager mode as well."""
    global _bridge
    if _bridge is False:
        _bridge = None
        path = os.path.join(os.path.dirname(LIB_PATH), "caster_gvp_torch.so")
        if os.environ.get("CGVP_BRIDGE", "1") != "0" and os.path.exists(path) and not os.environ.get("CGVP_LIB_PATH"):
            import importlib.util
            lib()                                    # libcaster_gvp.so first (version check; the bridge links against it)
            spec = importlib.util.spec_from_file_location("caster_gvp_torch", path)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            if mod.abi_version() != ABI_VERSION:          # the header version the bridge was COMPILED against
                raise HipLibraryError(f"{path} was built against C ABI {mod.abi_version()}, binding expects {ABI_VERSION}; rebuild")
            if mod.library_abi_version() != ABI_VERSION:  # ... and what the library it linked to reports at run time
                raise HipLibraryError(f"{path} is linked to a libcaster_gvp.so of ABI {mod.library_abi_version()}, "
                                      f"binding expects {ABI_VERSION}; rebuild")
            _bridge = mod
    return _bridge


def check(rc, what):
    if rc == 0:
        return
    if rc == -2:
        raise NotImplementedError(
            f"{what}: dimensions outside the compiled CASTER-DTA(s,v) configuration "
            "(node (17,3)/hidden (16,4)/edge (32,1)/out 64, one-hot widths {0,20,21}/{0,1})")
    if rc < 0:
        raise ValueError(f"{what}: bad argument (code {rc})")
    raise RuntimeError(f"{what}: HIP launch failed with hipError_t {rc}")
