"""Parameter arena: keep a module's nn.Parameters as views into ONE flat fp32
buffer laid out in state_dict order (include/caster_gvp.h, "PARAMETER ARENA").

The kernels read every weight at a compile-time offset from a single base
pointer; the optimizer, `load_state_dict`, checkpoints and `state_dict()` keep
working on the individual Parameters because those ARE the arena (views).  If
something re-materialises the parameters (``.to(device)``, ``.double()``,
manual ``p.data = ...``) the views are detected as stale on the next call and the
arena is rebuilt from the current values.
"""
from __future__ import annotations

import torch


class ParamArena:
    def __init__(self, params):
        """`params`: list of nn.Parameter in arena order (zero-size ones skipped)."""
        self.params = [p for p in params if p.numel() > 0]
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += p.numel()
        self.total = off
        self.flat = None

    def _is_current(self):
        f = self.flat
        if f is None:
            return False
        base = f.data_ptr()
        for p, off in zip(self.params, self.offsets):
            if p.data_ptr() != base + 4 * off:
                return False
        return True

    @torch.no_grad()
    def rebuild(self):
        p0 = self.params[0]
        flat = torch.empty(self.total, dtype=p0.dtype, device=p0.device)
        for p, off in zip(self.params, self.offsets):
            n = p.numel()
            flat[off:off + n].copy_(p.detach().reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
        self.flat = flat
        return flat

    def buffer(self):
        """The flat fp32 buffer, rebuilt first if any parameter left it."""
        if not self._is_current():
            self.rebuild()
        return self.flat

    def split(self, flat_like):
        """Views of a same-layout flat tensor (e.g. the gradient arena), one per parameter."""
        return [flat_like[off:off + p.numel()].view(p.shape) for p, off in zip(self.params, self.offsets)]


_GVP_KEYS = ("wh.weight", "ws.weight", "ws.bias", "wv.weight", "wsv.weight", "wsv.bias")
_LN_KEYS = ("scalar_norm.weight", "scalar_norm.bias")


def lba_param_keys(num_convs):
    """state_dict keys of VectorProteinGNN_LBAModel in arena order
    (include/caster_gvp.h; zero-size dummy_params are not part of the arena)."""
    keys = [f"gvp_node.0.{k}" for k in _GVP_KEYS] + [f"gvp_node.1.{k}" for k in _LN_KEYS]
    keys += [f"gvp_edge.0.{k}" for k in _GVP_KEYS] + [f"gvp_edge.1.{k}" for k in _LN_KEYS]
    for l in range(num_convs):
        for m in range(3):
            keys += [f"conv_list.{l}.conv.message_func.{m}.{k}" for k in _GVP_KEYS]
        for n in range(2):
            keys += [f"conv_list.{l}.norm.{n}.{k}" for k in _LN_KEYS]
        for f in range(2):
            keys += [f"conv_list.{l}.ff_func.{f}.{k}" for k in _GVP_KEYS]
    keys += [f"gvp_norm_before_scalar.{k}" for k in _LN_KEYS]
    keys += [f"gvp_to_scalar.{k}" for k in _GVP_KEYS[:3]]
    return keys


def flatten_state(state, num_convs, device=None):
    """Flat fp32 arena from a {key: tensor} mapping (tests / checkpoint tools)."""
    parts = [state[k].detach().reshape(-1).float() for k in lba_param_keys(num_convs)]
    flat = torch.cat(parts)
    return flat.to(device) if device is not None else flat


class FusedLeaf:
    """ONE trainable leaf for a module tree (opt-in `fuse_parameters()` of the two encoders).

    PyTorch's autograd engine spends 3-4 us of host time per parameter leaf in every backward pass (one AccumulateGrad
    node each); the kernels already write all weight gradients into one flat buffer.  `FusedLeaf` replaces the listed
    nn.Parameters of `root` by a single `nn.Parameter` (`root.arena`, their values end to end in the given order):

      * the per-module attributes (`lin.weight`, ...) stay readable: plain tensor views of the arena's storage, re-seated
        whenever the arena is re-materialised (`.to()`, `.float()`);
      * `state_dict()` / `load_state_dict(strict=True)` keep the per-tensor keys and shapes (hooks), so checkpoints are
        interchangeable with the unfused model and with the reference;
      * `named_parameters()` lists `arena` instead: build the optimizer AFTER fusing.  Cannot be undone.
    """

    def __init__(self, root, keys):
        """`keys`: dotted attribute paths below `root`, in arena order (zero-size parameters are skipped)."""
        self.root = root
        named = dict(root.named_parameters())
        self.items = [(k, tuple(named[k].shape), named[k].numel()) for k in keys if named[k].numel() > 0]
        flat = torch.cat([named[k].detach().reshape(-1) for k, _, _ in self.items])
        for k, _, _ in self.items:
            mod, name = self.owner(k)
            del mod._parameters[name]
        root.arena = torch.nn.Parameter(flat)
        self.seat()
        root._register_state_dict_hook(self._state_dict_hook)
        root._register_load_state_dict_pre_hook(self._load_hook)

    def owner(self, key):
        mod = self.root
        *path, name = key.split(".")
        for part in path:
            mod = mod[int(part)] if part.isdigit() else getattr(mod, part)
        return mod, name

    def views(self, flat):
        off = 0
        for k, shape, n in self.items:
            yield k, flat[off:off + n].view(shape)
            off += n

    def seat(self):
        for k, v in self.views(self.root.arena.detach()):
            mod, name = self.owner(k)
            object.__setattr__(mod, name, v)

    def _state_dict_hook(self, module, state_dict, prefix, local_metadata):
        flat = state_dict.pop(prefix + "arena")
        for k, v in self.views(flat):
            state_dict[prefix + k] = v

    def _load_hook(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        if prefix + "arena" in state_dict:
            return
        cur = self.root.arena.detach()
        parts = []
        for k, v in self.views(cur):
            t = state_dict.pop(prefix + k, None)
            if t is None:
                if strict:
                    missing_keys.append(prefix + k)
                parts.append(v.reshape(-1))
            elif tuple(t.shape) != tuple(v.shape):
                error_msgs.append(f"size mismatch for {prefix + k}: copying a param with shape {tuple(t.shape)} from "
                                  f"checkpoint, the shape in current model is {tuple(v.shape)}.")
                parts.append(v.reshape(-1))
            else:
                parts.append(t.detach().reshape(-1).to(cur))
        state_dict[prefix + "arena"] = torch.cat(parts)
