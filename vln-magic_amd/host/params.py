"""Flat parameter storage.

Every parameter of a model lives in ONE contiguous fp32 buffer (`flat`), with congruent buffers for
gradients, Adam moments and (bf16 mode) the bf16 shadow weights the MFMA kernels read.  nn.Parameters
are views into `flat`, named exactly as the reference's checkpoint keys (SURVEY App. A.4;
train_r2r_magic.py:189-208), so state_dict()/load_state_dict()/named_parameters() behave as for an
ordinary module while the optimizer and the gradient all-reduce see single large buffers.
Layout: [weight-decay group | no-decay group], the split following optim/misc.py:13-22 by name.
"""
import math
import weakref

import torch
import torch.nn as nn

from . import lanes

_LIVE = weakref.WeakSet()          # trainable stores with a bf16 shadow: a foreign optimizer's step must mark them stale
_HOOKED = [False]


def _install_optimizer_hook():
    """ANY torch.optim.Optimizer.step() (the reference's own AdamW, pretrain_src/optim/adamw.py, writes `p.data` -- invisible to
    tensor version counters) marks the bf16 shadow weights of every trainable store stale, so a forward that follows an
    external optimizer step re-casts them whatever ran in between (validation forwards, logging forwards).  FusedAdamW is not a
    torch Optimizer: it rewrites the shadow itself."""
    if _HOOKED[0]:
        return
    try:
        from torch.optim.optimizer import register_optimizer_step_post_hook
    except ImportError:          # old torch: ensure_grads()' conservative invalidation still covers backward -> step -> forward
        _HOOKED[0] = True
        return

    def _stale(optimizer, args, kwargs):
        for s in list(_LIVE):
            s.shadow_clean = False
    register_optimizer_step_post_hook(_stale)
    _HOOKED[0] = True

ALIGN = 64           # elements; keeps every tensor 256-byte aligned in fp32 and 128-byte in bf16
NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")      # optim/misc.py:14


def is_no_decay(name):
    return any(nd in name for nd in NO_DECAY)


def _attach(root, dotted, param):
    mod = root
    parts = dotted.split(".")
    for p in parts[:-1]:
        if not hasattr(mod, p) or not isinstance(getattr(mod, p), nn.Module):
            mod.add_module(p, nn.Module())
        mod = getattr(mod, p)
    mod.register_parameter(parts[-1], param)


class ParamStore:
    def __init__(self, specs, device, compute_dtype, init_std=0.02, seed=0, requires_grad=True):
        """specs: ordered list of (name, shape, kind) with kind in {'normal','zeros','ones'}."""
        self.device, self.compute_dtype = torch.device(device), compute_dtype
        self.specs = list(specs)
        order = [s for s in self.specs if not is_no_decay(s[0])] + [s for s in self.specs if is_no_decay(s[0])]
        self.order = order
        self.offsets, off = {}, 0
        self.n_decay = None
        for name, shape, _ in order:
            if self.n_decay is None and is_no_decay(name):
                self.n_decay = off
            n = int(math.prod(shape))
            self.offsets[name] = (off, n, tuple(shape))
            off += (n + ALIGN - 1) // ALIGN * ALIGN
        if self.n_decay is None:
            self.n_decay = off
        self.total = off
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self.requires_grad = requires_grad
        if requires_grad:
            self.grad = torch.zeros_like(self.flat)
            self.m = torch.zeros_like(self.flat)
            self.v = torch.zeros_like(self.flat)
        else:
            self.grad = self.m = self.v = None
        self.lane_grads = {}           # lane (>= 1) -> a second flat gradient buffer (host/lanes.py); summed into `grad` by merge_lanes()
        self.lane_dirty = False
        self.half = compute_dtype in (torch.bfloat16, torch.float16)      # 16-bit compute: the MFMA kernels read a shadow copy in that type
        self.shadow = torch.zeros(self.total, dtype=compute_dtype, device=self.device) if self.half else self.flat
        self.shadow_clean = False
        # transposed bf16 shadow (W^T at the same flat offset) of the matrices the backward row-block kernel reads (csrc/encbwd.hip);
        # spans register themselves through t_span(); refreshed right after the ordinary shadow
        self.shadow_t = None
        self.t_spans = {}
        self.shadow_t_clean = False
        # the same for matrices a kernel reads in MFMA-fragment order (csrc/chain.hip): f_span()
        self.shadow_f = None
        self.f_spans = {}
        self.shadow_f_clean = False
        self.shadow_tf = None          # ... and for W^T in fragment order (csrc/encbwd.hip); f_spans: offset -> (rows, cols, 1 = W | 2 = W^T)
        if requires_grad and self.half:
            _LIVE.add(self)
            _install_optimizer_hook()
        gen = torch.Generator().manual_seed(seed)
        for name, shape, kind in self.specs:
            v = self.master(name)
            if kind == "normal":
                v.copy_(torch.randn(shape, generator=gen) * init_std)
            elif kind == "ones":
                v.fill_(1.0)
        self.step = 0

    def checkpoint_like_(self, seed, bias_std=0.02, ln_std=0.05):
        """Small parameters as a trained checkpoint has them rather than as a fresh module does: biases ~ N(0, bias_std), LayerNorm gains
        1 + N(0, ln_std) -- the recipe of oracle/parity_probe.oracle_models.  The reference never trains from zero biases (its loop starts from METER's
        RoBERTa / cross-modal weights, pretrain_src/train_r2r_magic.py:183-209), and with them a LayerNorm fed by a Linear over an all-zero
        input row (padding, eps 1e-12) is LayerNorm(0): rstd = 1e6 and a gradient norm of 1e5 that the clip then turns into a 1e-5 update."""
        gen = torch.Generator().manual_seed(int(seed))
        for name, shape, kind in self.specs:
            v = self.master(name)
            if name.endswith("bias"):
                v.copy_(torch.randn(shape, generator=gen) * bias_std)
            elif kind == "ones":
                v.add_((torch.randn(shape, generator=gen) * ln_std).to(v.device))
        self.shadow_clean = False
        return self

    # ---- views ------------------------------------------------------------------------------
    def _view(self, buf, name):
        off, n, shape = self.offsets[name]
        return buf[off:off + n].view(shape)

    def master(self, name):
        return self._view(self.flat, name)

    def w(self, name):
        """compute-dtype view (bf16 shadow in bf16 mode, the fp32 master otherwise)"""
        return self._view(self.shadow, name)

    def _gbuf(self):
        """the flat gradient buffer of the current lane (host/lanes.py; lane 0 = `grad`)"""
        k = lanes.cur
        if k == 0:
            return self.grad
        b = self.lane_grads.get(k)
        if b is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("ParamStore: a gradient lane must be allocated before a graph capture (ensure_lanes)")
            b = self.lane_grads[k] = torch.zeros_like(self.grad)
        self.lane_dirty = True
        return b

    def ensure_lanes(self, n):
        """allocate the gradient buffers of lanes 1 .. n-1"""
        if self.requires_grad:
            for k in range(1, n):
                if k not in self.lane_grads:
                    self.lane_grads[k] = torch.zeros_like(self.grad)

    def merge_lanes(self):
        """grad += every lane's buffer (lane order), the lane buffers zeroed for the next pass; the caller has ordered the current stream behind the
        lanes' streams (lanes.join)"""
        if self.lane_dirty:
            for k in sorted(self.lane_grads):
                b = self.lane_grads[k]
                self.grad.add_(b)
                b.zero_()
            self.lane_dirty = False

    def g(self, name):
        return self._view(self._gbuf(), name)

    def w_span(self, first, rows, cols):
        """[rows, cols] compute-dtype view starting at `first` spanning consecutive tensors (fused QKV)."""
        off = self.offsets[first][0]
        return self.shadow[off:off + rows * cols].view(rows, cols)

    def t_span(self, first, rows, cols):
        """[cols, rows] bf16 view = transpose of the [rows, cols] span starting at `first`; kept current by sync_shadow / FusedAdamW"""
        off = self.offsets[first][0]
        if self.shadow_t is None:
            self.shadow_t = torch.zeros(self.total, dtype=self.compute_dtype, device=self.device)
        if off not in self.t_spans:
            self.t_spans[off] = (rows, cols)
            self.shadow_t_clean = False
            self._t_arrays = None
            if self.shadow_clean and self.device.type == "cuda":      # registered after the shadow was last refreshed: fill it in now
                self.sync_shadow_t(force=True)
        return self.shadow_t[off:off + rows * cols].view(cols, rows)

    def _layout_span(self, first, rows, cols, flag):
        off = self.offsets[first][0]
        if self.shadow_f is None:
            self.shadow_f = torch.zeros(self.total, dtype=self.compute_dtype, device=self.device)
            self.shadow_tf = torch.zeros(self.total, dtype=self.compute_dtype, device=self.device)
        have = self.f_spans.get(off)
        if have is None or not (have[2] & flag):
            assert have is None or have[:2] == (rows, cols), "one span, two shapes"
            self.f_spans[off] = (rows, cols, (have[2] if have else 0) | flag)
            self.shadow_f_clean = False
            self._f_arrays = None
            if self.shadow_clean and self.device.type == "cuda":      # registered after the shadow was last refreshed: fill it in now
                self.sync_shadow_f(force=True)
        return off

    def f_span(self, first, rows, cols):
        """the [rows, cols] span starting at `first` in MFMA-FRAGMENT ORDER (every 16 x 32 fragment's 64 lane operands contiguous: csrc/chain.hip
        magic_pack_frag_spans), as a flat compute-dtype view; kept current by sync_shadow / FusedAdamW"""
        off = self._layout_span(first, rows, cols, 1)
        return self.shadow_f[off:off + rows * cols]

    def tf_span(self, first, rows, cols):
        """W^T ([cols, rows]) of the [rows, cols] span starting at `first`, in fragment order (flat view; csrc/encbwd.hip reads it)"""
        off = self._layout_span(first, rows, cols, 2)
        return self.shadow_tf[off:off + rows * cols]

    def sync_shadow_f(self, force=False):
        """ONE launch: every registered span of the (clean) shadow -> W and / or W^T in fragment order"""
        if not self.f_spans or (self.shadow_f_clean and not force):
            return
        import ctypes as C
        from . import lib as L
        if getattr(self, "_f_arrays", None) is None:
            offs = sorted(self.f_spans)
            n = len(offs)
            self._f_arrays = (n, (C.c_longlong * n)(*offs), (C.c_int * n)(*[self.f_spans[o][0] for o in offs]),
                              (C.c_int * n)(*[self.f_spans[o][1] for o in offs]), (C.c_int * n)(*[self.f_spans[o][2] for o in offs]))
        n, a_off, a_rows, a_cols, a_flags = self._f_arrays
        L.call("magic_layout_spans", L.P(self.shadow), L.P(self.shadow_f), L.P(self.shadow_tf), n, C.addressof(a_off), C.addressof(a_rows),
               C.addressof(a_cols), C.addressof(a_flags), L.stream())
        self.shadow_f_clean = True

    def sync_shadow_t(self, force=False):
        """one launch: every registered span of the (clean) shadow -> its transpose"""
        if not self.t_spans or (self.shadow_t_clean and not force):
            return
        import ctypes as C
        from . import lib as L
        if getattr(self, "_t_arrays", None) is None:
            offs = sorted(self.t_spans)
            n = len(offs)
            self._t_arrays = (n, (C.c_longlong * n)(*offs), (C.c_int * n)(*[self.t_spans[o][0] for o in offs]),
                              (C.c_int * n)(*[self.t_spans[o][1] for o in offs]))
        n, a_off, a_rows, a_cols = self._t_arrays
        L.call("magic_transpose_spans", L.P(self.shadow), L.P(self.shadow_t), n, C.addressof(a_off), C.addressof(a_rows), C.addressof(a_cols), L.stream())
        self.shadow_t_clean = True

    def master_span(self, first, n):
        off = self.offsets[first][0]
        return self.flat[off:off + n]

    def g_span(self, first, n):
        off = self.offsets[first][0]
        return self._gbuf()[off:off + n]

    def first_offset(self, prefixes, decay):
        """offset of the first tensor (in storage order, within the weight-decay or the no-decay group) whose name starts with one
        of `prefixes`; the group's end if none does"""
        for name, _, _ in self.order:
            if is_no_decay(name) != decay and name.startswith(tuple(prefixes)):
                return self.offsets[name][0]
        return self.n_decay if decay else self.total

    def ranges_of(self, prefixes):
        """sorted, merged [lo, hi) element ranges of the flat buffers covering every tensor whose name starts with one of `prefixes`
        (alignment gaps between neighbouring matches are absorbed; a range never spans the weight-decay | no-decay boundary)"""
        hit = sorted((self.offsets[n][0], self.offsets[n][0] + self.offsets[n][1]) for n, _, _ in self.order if n.startswith(tuple(prefixes)))
        nxt = {}                      # end of a tensor's padded slot = start of the next tensor in storage order (or the group's end)
        starts = sorted(o for o, _, _ in self.offsets.values()) + [self.total]
        for a, b in zip(starts[:-1], starts[1:]):
            nxt[a] = b
        out = []
        for lo, hi in hit:
            hi_pad = min(nxt[lo], self.n_decay) if lo < self.n_decay else nxt[lo]
            if out and out[-1][1] == lo:
                out[-1][1] = hi_pad
            else:
                out.append([lo, hi_pad])
        return [(a, b) for a, b in out]

    def contiguous(self, names):
        """True if the tensors are laid out back to back (no alignment gap)."""
        for a, b in zip(names[:-1], names[1:]):
            oa, na, _ = self.offsets[a]
            if self.offsets[b][0] != oa + na:
                return False
        return True

    # ---- module integration -----------------------------------------------------------------
    def attach_to(self, module):
        self._params = []
        for name, _, _ in self.specs:
            p = nn.Parameter(self.master(name), requires_grad=self.requires_grad)
            if self.requires_grad:
                p.grad = self.g(name)
            _attach(module, name, p)
            self._params.append((name, p))

    def ensure_grads(self):
        """Called at the start of every explicit backward.  `optimizer.zero_grad()` of an unmodified training loop sets
        `.grad = None` (torch >= 2.0 default; the reference's torch 1.9 zeroed in place): None means zero, so the flat
        gradient buffer is cleared and every parameter's `.grad` is pointed back at its slice of it."""
        ps = getattr(self, "_params", None)
        if not self.requires_grad or not ps:
            return
        # a backward means an optimizer step follows; if that optimizer is not FusedAdamW (which refreshes the bf16 shadow
        # itself and sets the flag again) the next forward must re-cast the weights
        self.shadow_clean = False
        if ps[0][1].grad is None or ps[-1][1].grad is None:
            self.grad.zero_()
            for name, p in ps:
                p.grad = self.g(name)

    def sync_shadow(self, force=False):
        if not self.half:
            return
        if force or not self.shadow_clean:
            from . import lib as L
            L.call("magic_cast", L.dt(self.compute_dtype), 1, self.total, L.P(self.flat), L.P(self.shadow), L.stream())
            self.shadow_clean = True
            self.shadow_t_clean = False
            self.shadow_f_clean = False
        self.sync_shadow_t()
        self.sync_shadow_f()

    def zero_grad(self):
        self.grad.zero_()
        for b in self.lane_grads.values():       # always: a replayed lane graph writes its buffer without touching a handle (lane_dirty may be unset), and a
            b.zero_()                            # pass that raised before its lanes were merged leaves them filled
        self.lane_dirty = False
