"""Host-side mirror of the reference's conditional denoiser ``Unet``
(/root/reference/ddpm.py:286-451) over the gfx950 HIP kernels.

* Same constructor signature, same attributes read by ``GaussianDiffusion`` (``channels``,
  ``out_dim``, ``self_condition``, ``random_or_learned_sinusoidal_cond``, ``downsample_factor``),
  same ``forward(x, cond_img, time, x_self_cond=None)`` contract (NCHW fp32 in / out), and a
  ``state_dict`` with exactly the reference's parameter names, so checkpoints
  written by the reference's ``Trainer.save`` load by name.
* The arithmetic runs entirely in ``csrc/liblocaldiff_hip.so`` through ctypes: the forward is a
  *plan* -- a list of C-ABI calls with pre-built argument structs over static NHWC buffers --
  built once per (batch, H, W) and replayed, eagerly or from a captured HIP graph.  PyTorch only
  owns device memory and the stream.  Without the library this module raises.

Fusions relative to the reference's op-per-module execution are listed in DESIGN.md; numerically
the plan computes the same function (fp32 mode agrees with the reference to ~1e-5 per forward).
"""
import ctypes as C
import math

import numpy as np
import torch
from torch import nn

from . import _cabi as cabi
from .pool import check_declared, intervals_from_declared, peak_live, place_intervals
from .tuning import Tuning
from .weights import UnetConfig, unet_param_shapes

_TORCH_DT = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}
_LOWP = ("bf16", "fp16")          # 16-bit storage: fused linear attention, MFMA stem, folded conv_fusion


def _register(root: nn.Module, dotted: str, tensor: torch.Tensor):
    """Create bare container modules along ``dotted`` and register the leaf parameter, so that
    ``state_dict()`` keys equal the reference's (e.g. ``downs.0.0.block1.proj.weight``)."""
    parts = dotted.split(".")
    mod = root
    for name in parts[:-1]:
        if name not in mod._modules:
            mod.add_module(name, nn.Module())
        mod = mod._modules[name]
    mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


class Unet(nn.Module):
    def __init__(self, dim, init_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8), channels=1,
                 self_condition=False, cond_img=True, resnet_block_groups=8, learned_variance=False,
                 learned_sinusoidal_cond=False, random_fourier_features=False,
                 learned_sinusoidal_dim=16, sinusoidal_pos_emb_theta=10000, attn_dim_head=32,
                 attn_heads=4, full_attn=(False, False, False, True), flash_attn=False, mode="mri",
                 compute_dtype="fp32", tuning=None):
        super().__init__()
        # The reference's remaining constructor options (ddpm.py:294-300; no shipped caller sets them, test.py:117-129):
        #   learned_variance      -> only the default out_dim doubles (:394); Unet.forward is unchanged;
        #   learned_sinusoidal_cond / random_fourier_features -> time_mlp.0 is RandomOrLearnedSinusoidalPosEmb (:151-165, a
        #       `weights` parameter of learned_sinusoidal_dim / 2 frequencies) and time_mlp.1 takes dim + 1 features;
        #   self_condition        -> accepted like the reference accepts it; its forward then concatenates x_self_cond into an
        #       init_conv that was built for `channels` inputs (:315-319, :406-408) and fails there -- mirrored in forward().
        if (learned_sinusoidal_cond or random_fourier_features) and (learned_sinusoidal_dim % 2 or not 2 <= learned_sinusoidal_dim <= 256):
            raise ValueError("learned_sinusoidal_dim must be even (ddpm.py:156) and at most 256")
        if dim != 32:
            raise ValueError("dim must be 32: the conditioning encoder hard-codes its widths "
                             "(unet_model.py:100) and is concatenated with the 8*dim bottleneck")
        if attn_dim_head != 32:
            raise ValueError("attn_dim_head must be 32 (MFMA tile of the attention kernels)")
        if compute_dtype not in _TORCH_DT:
            raise ValueError(f"compute_dtype {compute_dtype!r} (fp32 | bf16 | fp16)")
        full_attn = tuple(full_attn) if isinstance(full_attn, (tuple, list)) else (full_attn,) * len(dim_mults)
        assert len(full_attn) == len(dim_mults)
        init_dim = dim if init_dim is None else init_dim
        learned = bool(learned_sinusoidal_cond or random_fourier_features)
        default_out = channels * (2 if learned_variance else 1)                    # ddpm.py:394
        self.cfg = UnetConfig(dim=dim, init_dim=init_dim, out_dim=default_out if out_dim is None else out_dim,
                              dim_mults=tuple(dim_mults), channels=channels,
                              resnet_block_groups=resnet_block_groups, attn_dim_head=attn_dim_head,
                              attn_heads=attn_heads, full_attn=full_attn, mode=mode,
                              learned_sinusoidal_dim=int(learned_sinusoidal_dim) if learned else 0)
        self.mode = mode
        self.channels = channels
        self.out_dim = self.cfg.out_dim
        self.self_condition = bool(self_condition)
        self.cond_img = cond_img
        self.random_or_learned_sinusoidal_cond = learned
        self.theta = sinusoidal_pos_emb_theta
        self.compute_dtype = compute_dtype
        g = torch.Generator().manual_seed(0)
        shapes = unet_param_shapes(self.cfg)
        for name, shape in shapes.items():
            if name.endswith(".g"):
                t = torch.ones(shape)
            elif name.endswith(".weights"):
                t = torch.randn(shape, generator=g)                                # ddpm.py:158
            elif len(shape) == 1 and (".norm." in name or "convblock.1" in name or "convblock.4" in name
                                      or "identity.1" in name):
                t = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
            else:
                wshape = shape if name.endswith("weight") else shapes[name[:-4] + "weight"]
                bound = 1.0 / math.sqrt(int(np.prod(wshape[1:])))
                t = (torch.rand(shape, generator=g) * 2 - 1) * bound
            _register(self, name, t)
        self._packed = None
        self._packed_key = None
        self._plans = {}
        # every performance knob in one object (tuning.py); default: compiled-in defaults + LD_* environment overrides
        self.tuning = tuning if tuning is not None else Tuning.from_env()
        # two-term convolution weights (W = hi + lo in 16-bit storage, twice the matrix work) for the layers of the
        # first N resolution levels (0 = off; 2 = the full- and half-resolution layers, which carry 90 % of what
        # rounding the weights costs a sampling chain: DESIGN section 2).  See set_weight_split_levels().
        self.weight_split_levels = int(self.tuning.weight_split_levels)
        self._version = 0

    # ------------------------------------------------------------------ reference attributes
    @property
    def downsample_factor(self):
        return self.cfg.downsample_factor

    @property
    def device(self):
        return next(self.parameters()).device

    def set_compute_dtype(self, name):
        if name not in _TORCH_DT:
            raise ValueError(name)
        if name != self.compute_dtype:
            self.compute_dtype = name
            self.invalidate()

    def set_weight_split_levels(self, n):
        """Two-term weights for the convolutions of the first ``n`` resolution levels (16-bit storage)."""
        if int(n) != self.weight_split_levels:
            self.weight_split_levels = int(n)
            self.invalidate()

    def _layer_level(self, name):
        """Resolution level a parameter's layer works at: 0 = full resolution ... len(dim_mults) - 1 = coarsest."""
        n = len(self.cfg.dim_mults)
        if name.startswith("downs."):
            return int(name.split(".")[1])
        if name.startswith("ups."):
            return n - 1 - int(name.split(".")[1])
        if name.startswith("final_res_block"):
            return 0
        return n - 1

    def invalidate(self):
        """Drop packed weights and plans (call after changing parameters in place)."""
        self._packed, self._packed_key, self._plans = None, None, {}
        self._version = getattr(self, "_version", 0) + 1      # samplers drop graphs / sub-batch runners built on old plans

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self.invalidate()
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self.invalidate()
        return out

    # ------------------------------------------------------------------ weight packing
    def packed(self):
        """Kernel-layout weights for the current device/dtype (built lazily, cached)."""
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("the HIP denoiser needs its parameters on a GPU (model.to('cuda')); "
                               "there is no CPU execution path")
        key = (str(dev), self.compute_dtype, self.weight_split_levels)
        if self._packed is not None and self._packed_key == key:
            return self._packed
        lib = cabi.lib()
        self.tuning.apply_kernel_table(lib)
        st = torch.cuda.current_stream().cuda_stream
        dt, tdt = cabi.dtype_code(self.compute_dtype), _TORCH_DT[self.compute_dtype]
        sd = {k: v.detach().to(torch.float32).contiguous() for k, v in self.state_dict().items()}
        P = {"f32": sd, "w": {}, "g2": {}, "terms2": set()}

        def pack(name, ksize, scale_in=None, unshuffle=0):
            w = sd[name]
            cout, cin = w.shape[0], w.shape[1]
            terms = 2 if (self.compute_dtype in _LOWP and scale_in is None and not name.startswith("cond_model.")
                          and self._layer_level(name) < self.weight_split_levels) else 1
            out = torch.empty(terms * cout * cin * ksize * ksize, dtype=tdt, device=dev)
            cabi.check(lib.ld_pack_conv_weight_terms(w.data_ptr(), cabi.ptr(scale_in), out.data_ptr(), cout, cin,
                                                     ksize, unshuffle, dt, terms, st), "pack " + name)
            P["w"][name] = out
            if terms == 2:
                P["terms2"].add(out.data_ptr())
            if scale_in is not None:
                P.setdefault("keep", []).append(scale_in)

        for name, w in sd.items():
            if not name.endswith(".weight") or w.dim() != 4:
                continue
            k = w.shape[-1]
            base = name[:-len(".weight")]
            if k == 7 or w.shape[1] < 32:
                continue                                  # image convs stay OIHW fp32
            if base == "final_conv":
                continue
            if base.endswith(".to_qkv"):
                g = sd[base[:-len(".to_qkv")] + ".norm.g"].flatten()
                pack(name, 1, scale_in=(g * math.sqrt(g.numel())).contiguous())
            elif k == 1 and base.endswith(".3.1") and base.startswith("downs."):
                pack(name, 1, unshuffle=1)
            elif k == 1 and (base.endswith(".to_out.0")):
                continue                                  # folded per batch with ctx (ld_linattn_fold)
            else:
                pack(name, k)
        P["wq"], P["wkv"], P["kshift"], P["qshift"] = {}, {}, {}, {}
        if self.compute_dtype in _LOWP:
            hid, heads = self.cfg.hidden, self.cfg.attn_heads
            for name, w in sd.items():
                if not name.endswith(".to_qkv.weight") or (name[:-len(".to_qkv.weight")] + ".to_out.1.g") not in sd:
                    continue
                base = name[:-len(".to_qkv.weight")]
                g = sd[base + ".norm.g"].flatten()
                scale = (g * math.sqrt(g.numel())).contiguous()
                c = w.shape[1]
                if c not in (32, 64, 128):
                    continue

                terms = 2 if (c in (32, 64) and heads == 4 and self._layer_level(name) < self.weight_split_levels) else 1

                def pack_rows(rows, terms=terms):
                    out = torch.empty(terms * rows.numel(), dtype=tdt, device=dev)
                    rows = rows.contiguous()
                    cabi.check(lib.ld_pack_conv_weight_terms(rows.data_ptr(), scale.data_ptr(), out.data_ptr(), rows.shape[0],
                                                             c, 1, 0, dt, terms, st), "pack " + name)
                    P.setdefault("keep", []).append(rows)
                    return out
                P.setdefault("attn_terms", {})[base] = terms
                P["wq"][base] = pack_rows(w[:hid])
                per_head = []
                for h in range(heads):
                    kv = torch.cat([w[hid + 32 * h: hid + 32 * h + 32], w[2 * hid + 32 * h: 2 * hid + 32 * h + 32]], 0)
                    per_head.append(pack_rows(kv))
                P["wkv"][base] = torch.cat(per_head).contiguous()
                # softmax_n(k) shift for the single-sweep kernel: the RMS-normalised pixel has unit 2-norm, so
                # |k_d| <= ||W_k[d,:] * g * sqrt(C)||_2 (of the storage-rounded weights, + 1 % slack).  Above 40
                # exp(k - bound) could underflow in fp32: fall back to the exact two-sweep maximum (None).
                wk = (w[hid:2 * hid, :, 0, 0].float() * scale[None, :]).to(tdt).float()
                bound = wk.norm(dim=1) * 1.01
                # fp16 stores P = exp(k - bound) * 2^10 with a normal range of 2^-14: a true maximum 2*bound below
                # the bound must stay inside it, so the single sweep is only used for bounds <= 8 there
                kmax_ok = 40.0 if self.compute_dtype == "bf16" else 8.0
                P["kshift"][base] = bound.contiguous() if float(bound.max()) <= kmax_ok else None
                wqn = (w[:hid, :, 0, 0].float() * scale[None, :]).to(tdt).float().norm(dim=1) * 1.01
                qb = wqn.reshape(heads, 32).amax(dim=1)          # softmax_d(q) shift per head (same bound, max over d)
                P["qshift"][base] = qb.contiguous() if float(qb.max()) <= 40.0 else None
                P["keep"].append(scale)
        if self.compute_dtype in _LOWP and "conv_fusion.res_conv.weight" in sd:
            # conv_fusion sees cat(trunk, conditioning features) (ddpm.py:434-436) and the second half does not change
            # between reverse steps: its weights are packed per half so that the conditioning half of block1.proj and
            # res_conv can be evaluated once per sample (16-bit storage only: fp32 keeps the reference's summation order)
            for nm, k in (("conv_fusion.block1.proj.weight", 3), ("conv_fusion.res_conv.weight", 1)):
                wfull = sd[nm]
                half = wfull.shape[1] // 2
                for tag, part in (("#x", wfull[:, :half]), ("#c", wfull[:, half:])):
                    part = part.contiguous()
                    out = torch.empty(part.numel(), dtype=tdt, device=dev)
                    cabi.check(lib.ld_pack_conv_weight(part.data_ptr(), None, out.data_ptr(), part.shape[0], part.shape[1],
                                                       k, 0, dt, st), "pack " + nm + tag)
                    P["w"][nm + tag] = out
                    P.setdefault("keep", []).append(part)
        if self.compute_dtype in _LOWP and self.cfg.init_dim == 32:
            wi = sd["init_conv.weight"].contiguous()
            stem = torch.empty(int(lib.ld_stem_packed_bytes()), dtype=torch.uint8, device=dev)
            cabi.check(lib.ld_pack_stem_weight(wi.data_ptr(), stem.data_ptr(), wi.shape[1], st), "pack init_conv")
            P["stem"] = stem
            P.setdefault("keep", []).append(wi)
        for name, w in sd.items():
            if name.endswith(".to_out.1.g"):
                g = w.flatten()
                P["g2"][name] = (g * math.sqrt(g.numel())).contiguous()
        half = self.cfg.dim // 2
        step = math.log(self.theta) / (half - 1)
        P["freqs"] = torch.exp(torch.arange(half) * -step).to(dev)          # ddpm.py:145-146
        torch.cuda.current_stream().synchronize()
        self._packed, self._packed_key = P, key
        return P

    # ------------------------------------------------------------------ plans
    def plan(self, B, H, W, table_T=None, instance=0):
        """``instance`` > 0: further plans of the same shape with their own buffers (concurrent sub-batches)."""
        f = self.downsample_factor
        assert H % f == 0 and W % f == 0, \
            f"your input dimensions {(H, W)} need to be divisible by {f}, given the unet"   # ddpm.py:405
        # (the plan-builder fields of the tuning are part of the key: assigning another Tuning after the first plan must
        #  not silently keep plans built under the old one -- ADVICE r4)
        tn = self.tuning
        key = (B, H, W, table_T, self.compute_dtype, instance, self.weight_split_levels,
               tn.separate_act, tn.sep_act_max_px, tn.sep_act_min_c, tn.fusion_fold, tn.side_res_conv, tn.side_res_conv_max_px, tn.side_res_conv_px, tn.fused_step_begin, tn.linattn_chunk_px, tn.buffer_reuse, tn.recompute_stem, tn.pool_by_size, tn.pool_verify)
        if key not in self._plans:
            self._plans[key] = _Plan(self, B, H, W, table_T)
        return self._plans[key]

    @torch.no_grad()
    def forward(self, x, cond_img, time, x_self_cond=None):
        """ddpm.py:404-451.  x [B,C,H,W], cond_img [B,Cc,H,W], time int64 [B] -> [B,out_dim,H,W]."""
        B, _, H, W = x.shape
        if self.self_condition:
            # ddpm.py:406-408 concatenates x_self_cond (zeros by default) in front of x, and :413 feeds the 2 * channels result to an
            # init_conv built for `channels` inputs (:315-319): the reference raises there, whatever the inputs.  Same here.
            c = self.cfg.channels
            raise RuntimeError(f"Given groups=1, weight of size [{self.cfg.init_dim}, {c}, 7, 7], expected input[{B}, {2 * c}, {H}, {W}] to have "
                               f"{c} channels, but got {2 * c} channels instead (self_condition=True: ddpm.py:406-413)")
        p = self.plan(B, H, W)
        st = torch.cuda.current_stream().cuda_stream
        p.x_in.copy_(x.to(torch.float32))
        p.cond_in.copy_(cond_img.to(torch.float32))
        p.times.copy_(time.to(torch.int32))
        p.run_time(st)
        p.run_cond(st)
        p.run_main(st)
        return p.model_out.clone()


class _Plan:
    """Static buffers + the C-ABI call list of one denoiser evaluation at a fixed shape.

    table_T=None : per-batch timesteps in ``self.times`` (the general ``Unet.forward`` contract);
    table_T=T    : FiLM tables for t = 0..T-1 are precomputed and kernels index them through the
                   device step counter ``self.t_dev`` (sampler fast path, HIP-graph replayable).
    """

    def __init__(self, net: Unet, B, H, W, table_T=None):
        self.net, self.B, self.H, self.W, self.table_T = net, B, H, W, table_T
        self.lib = cabi.lib()
        self.dev = net.device
        self.dt = cabi.dtype_code(net.compute_dtype)
        self.tdt = _TORCH_DT[net.compute_dtype]
        self.esize = 4 if self.dt == cabi.LD_F32 else 2
        self.P = net.packed()
        self.tn = net.tuning
        self.f32 = self.P["f32"]
        cfg = net.cfg
        self.cfg = cfg
        self.keep = []
        self._track = None                         # buffers of the main trunk, while buffer_reuse collects them
        self.named = {}                            # oracle tap name -> NHWC buffer (parity tests)
        self.meta = {}                             # index in ops_main -> {what, family, bytes, flops}
        self.decl = {}                             # index in ops_main -> (tensors read, tensors written): what the launch DECLARES (pool liveness)
        self.live_out = []                         # tensors of the main trunk something reads AFTER the launch list (the sampler's fused final step)
        self.ops_time, self.ops_cond, self.ops_main = [], [], []
        self.nslot = 0
        self.stats = torch.zeros(160, B, cabi.STAT_STRIPES, 16, 2, dtype=torch.float64, device=self.dev)
        self.cond_slots = 16                       # slots [0,16) belong to the conditioning encoder
        self.kmax_arena = torch.zeros(8, B, cabi.STAT_STRIPES, cfg.hidden, dtype=torch.int32, device=self.dev)   # zeroed per forward
        self._kmax_cursor = 0
        self.x_in = torch.zeros(B, cfg.channels, H, W, dtype=torch.float32, device=self.dev)
        self.cond_in = torch.zeros(B, cfg.cond_in_channels, H, W, dtype=torch.float32, device=self.dev)
        self.model_out = torch.zeros(B, cfg.out_dim, H, W, dtype=torch.float32, device=self.dev)
        self.times = torch.zeros(B, dtype=torch.int32, device=self.dev)
        self.t_dev = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.rows = table_T if table_T else B
        self.temb = torch.zeros(self.rows, cfg.time_dim, dtype=torch.float32, device=self.dev)
        self.films = {}
        # table mode: every block's FiLM table is a column range of ONE [T, film_row] arena and the step's first launch
        # (ld_step_begin_film) copies the row of the current timestep to ``film_cur``: consumers read (scale, shift) from a
        # fixed address, without the dependent load of the step counter (see src())
        self._film_off, self.film_row = {}, 0
        self._build_time()
        self.film_rows = self.film_cur = None
        if table_T and self.film_row:
            self.film_rows = torch.zeros(table_T, self.film_row, dtype=torch.float32, device=self.dev)
            self.film_cur = torch.zeros(self.film_row, dtype=torch.float32, device=self.dev)
        self._slot_cursor = 0
        self.cond_feat = self._build_cond()
        self.named["cond_model"] = self.cond_feat
        self.fusion_const = None
        if "conv_fusion.block1.proj.weight#c" in self.P["w"] and not cfg.cond_early_exit and self.tn.fusion_fold:
            c = self.cond_feat.shape[-1]
            hh, ww = self.cond_feat.shape[1], self.cond_feat.shape[2]
            p1 = self.conv3(self.ops_cond, [self.src(self.cond_feat, c)], "conv_fusion.block1.proj", c, hh, ww,
                            weight=self.P["w"]["conv_fusion.block1.proj.weight#c"])
            p2 = self.conv1(self.ops_cond, [self.src(self.cond_feat, c)], self.P["w"]["conv_fusion.res_conv.weight#c"],
                            c, hh, ww, bias=self.f32["conv_fusion.res_conv.bias"], what="res_conv(cond) conv_fusion")
            self.fusion_const = (p1, p2, torch.zeros(c, dtype=torch.float32, device=self.dev))
        self._slot_cursor = self.cond_slots
        self._track = [] if (table_T and self.tn.buffer_reuse) else None
        self._build_main()
        if self._track is not None:
            self._pool_buffers()
            self._track = None
        self._slots_used = self._slot_cursor       # statistics slots [cond_slots, _slots_used) are zeroed per evaluation
        s_, km_ = self.stats[self.cond_slots:self._slots_used], self.kmax_arena[:self._kmax_cursor]
        self._begin_args = (s_.data_ptr(), s_.numel() * 8, km_.data_ptr() if km_.numel() else None, km_.numel() * 4)
        self._t_dev_ptr = self.t_dev.data_ptr()
        self._film_args = (None, 0, None)
        if table_T:
            st = torch.cuda.current_stream().cuda_stream
            for op in self.ops_time:
                op(st)
            if self.film_rows is not None:             # setup: the per-block tables become column ranges of the arena
                for film in self.films.values():
                    off = self._film_off[id(film)]
                    self.film_rows[:, off:off + film.shape[1]].copy_(film)
                self._film_args = (self.film_rows.data_ptr(), self.film_row, self.film_cur.data_ptr())
            torch.cuda.current_stream().synchronize()

    # ------------------------------------------------------------------ helpers
    def _kt(self, name):
        """An entry of the library's launch-routing table (the plan's family labels follow the dispatcher's rule)."""
        v = C.c_longlong()
        cabi.check(self.lib.ld_tuning_get(name.encode(), C.byref(v)), "tuning_get")
        return int(v.value)

    def _separate_act(self, px, c):
        """block1's GroupNorm + FiLM + SiLU as a separate pass (instead of block2's prologue): small, wide maps
        (Tuning.separate_act / sep_act_max_px / sep_act_min_c; DESIGN findings 5, 68)."""
        return self.tn.separate_act and px <= self.tn.sep_act_max_px and c >= self.tn.sep_act_min_c

    def buf(self, h, w, c):
        t = torch.empty(self.B, h, w, c, dtype=self.tdt, device=self.dev)
        if self._track is not None:
            self._track.append(t)
        return t

    def _tracked(self, t):
        """A per-evaluation scratch tensor of the main trunk that is not an activation (context partials, folded matrices):
        placed in the pool like the buffers of ``buf``."""
        if self._track is not None:
            self._track.append(t)
        return t

    def _pool_buffers(self):
        """Sampler plans: the activations of the main trunk share ONE pool, placed by liveness (first / last launch that
        touches a buffer), instead of one allocation per layer -- a 4-patch plan cycles through ~0.65 GB of distinct
        addresses per evaluation otherwise, 2.5x the Infinity Cache, and every output line is written back to HBM before its
        address is seen again.

        Liveness is DECLARED (round 6): every launch of ``ops_main`` carries the tensors it reads and writes (``self.decl``,
        filled by the builders), tensors read after the list are in ``self.live_out``.  The launch list is still analysed as
        it stands -- ctypes argument blocks scanned for pointers into the tracked buffers, raw launches for the tensors
        their closures capture -- because that is what PATCHES the pointers and re-points the tensors; the scan is also
        the cross-check of the declarations (pool.check_declared: a buffer a launch touches without declaring it, or
        declares without a patchable reference, raises).  After patching everything is scanned again, plain integers
        included: nothing may still point into an old allocation.  ``named`` taps of pooled buffers are dropped.
        ``Tuning.pool_verify`` (LD_POOL_VERIFY=1): every buffer is overwritten with 0xFF bytes (NaN in every storage type)
        right behind the launch of its last declared use, so an undeclared later read shows in the samples."""
        import bisect
        bufs = self._track or []
        if not bufs:
            return
        rng = [(t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()) for t in bufs]
        order = sorted(range(len(bufs)), key=lambda k: rng[k][0])
        starts = [rng[k][0] for k in order]
        n_ops = len(self.ops_main)

        def owner(ptr):
            j = bisect.bisect_right(starts, ptr) - 1
            if j >= 0 and rng[order[j]][0] <= ptr < rng[order[j]][1]:
                return order[j]
            return None

        def scan_struct(obj, hit):
            """every c_void_p field of a ctypes block (nested structs / arrays included) that points into a tracked buffer"""
            if isinstance(obj, C.Array):
                for e in obj:
                    if isinstance(e, (C.Structure, C.Array)):
                        scan_struct(e, hit)
                return
            for name, typ in obj._fields_:
                v = getattr(obj, name)
                if isinstance(v, (C.Structure, C.Array)):
                    scan_struct(v, hit)
                elif typ is cabi.vp and v:
                    k = owner(v)
                    if k is not None:
                        hit(obj, name, k, v - rng[k][0])

        def tensors(v, depth=0):
            if isinstance(v, torch.Tensor):
                yield v
            elif isinstance(v, (tuple, list)) and depth < 3:
                for e in v:
                    yield from tensors(e, depth + 1)
            elif isinstance(v, dict) and depth < 3:
                for e in v.values():
                    yield from tensors(e, depth + 1)

        def held(op):
            """what a launch closure holds: default arguments and closure cells"""
            for d in (getattr(op, "__defaults__", None) or ()):
                yield d
            for cell in (getattr(op, "__closure__", None) or ()):
                try:
                    yield cell.cell_contents
                except ValueError:
                    continue

        # ---- the reflection scan: per launch, the tracked buffers it references in a PATCHABLE form
        scanned, patches = [set() for _ in range(n_ops)], []
        for i, op in enumerate(self.ops_main):
            def hit(obj, name, k, delta, i=i):
                scanned[i].add(k)
                patches.append((obj, name, k, delta))
            for v in held(op):
                if hasattr(v, "_obj"):                       # C.byref(args) of a _call launch
                    scan_struct(v._obj, hit)
                for t in tensors(v):
                    k = owner(t.data_ptr()) if t.is_cuda else None
                    if k is not None:
                        scanned[i].add(k)
        # ---- the declarations, and their agreement with the scan
        def declared_set(ts):
            out = set()
            for t in ts:
                k = owner(t.data_ptr()) if (t is not None and t.is_cuda) else None
                if k is not None:
                    out.add(k)
            return out
        declared = [declared_set(self.decl.get(i, ((), ()))[0]) | declared_set(self.decl.get(i, ((), ()))[1]) for i in range(n_ops)]
        op_names = [self.meta.get(i, {}).get("what", "?") for i in range(n_ops)]
        buf_names = [f"#{k} {tuple(t.shape)}" for k, t in enumerate(bufs)]
        check_declared(declared, scanned, op_names, buf_names)
        live_out = declared_set(self.live_out)
        for name, v in self.__dict__.items():              # cross-check: a tracked buffer reachable from a plan attribute is read
            if name in ("_track", "keep", "named", "decl", "live_out"):     # by something outside the launch list: it must be declared
                continue
            for t in tensors(v):
                k = owner(t.data_ptr()) if t.is_cuda else None
                if k is not None and k not in live_out:
                    raise RuntimeError(f"pool: plan attribute {name!r} reaches pooled buffer {buf_names[k]} that is not in live_out")
        first, last = intervals_from_declared(declared, len(bufs), live_out)
        size = [(r[1] - r[0] + 255) // 256 * 256 for r in rng]
        # placement (pool.py): largest buffers first, each at the lowest offset free of every placed buffer whose closed interval
        # meets its own -- the pool then equals the peak live set (by first use it was 9 % above it; at 8 patches per GPU the
        # working set sits on the edge of the cache, finding 108)
        offset, top = place_intervals(size, first, last, self.tn.pool_by_size)
        self._pool = torch.empty(top, dtype=torch.uint8, device=self.dev)
        base = self._pool.data_ptr()
        for obj, name, k, delta in patches:
            setattr(obj, name, base + offset[k] + delta)
        pooled = set()
        for k, t in enumerate(bufs):
            t.set_(self._pool.untyped_storage(), offset[k] // t.element_size(), t.size(), t.stride())
            assert t.data_ptr() == base + offset[k]
            if last[k] < n_ops:
                pooled.add(id(t))
        # ---- after patching: NOTHING a launch holds may still point into an old allocation -- pointer fields, tensors, and plain
        # integers (a data_ptr() taken at build time would have been missed by the scan above and is caught here)
        def stale(v):
            return isinstance(v, int) and not isinstance(v, bool) and owner(v) is not None
        cur = [0]

        def still(obj, name, k, delta):
            raise RuntimeError(f"pool: launch {cur[0]} ({op_names[cur[0]]}) field {name} still points into the old allocation of {buf_names[k]}")
        for i, op in enumerate(self.ops_main):
            cur[0] = i
            for v in held(op):
                if hasattr(v, "_obj"):
                    scan_struct(v._obj, still)
                vals = list(v) if isinstance(v, (tuple, list)) else (list(v.values()) if isinstance(v, dict) else [v])
                for e in vals:
                    if stale(e):
                        raise RuntimeError(f"pool: launch {i} ({op_names[i]}) holds the old address of {buf_names[owner(e)]} as a plain integer")
                for t in tensors(v):
                    if t.is_cuda and owner(t.data_ptr()) is not None:
                        raise RuntimeError(f"pool: launch {i} ({op_names[i]}) holds a tensor that was not re-pointed at the pool")
        self.named = {k: v for k, v in self.named.items() if id(v) not in pooled}
        peak = peak_live(size, first, last)
        self.pool_stats = dict(buffers=len(bufs), bytes_unshared=sum(size), bytes_pool=top, bytes_peak_live=peak)
        self.pool_intervals = [(first[k], last[k], offset[k], size[k]) for k in range(len(bufs))]
        if self.tn.pool_verify:
            dying = {}
            for k in range(len(bufs)):
                if 0 <= last[k] < n_ops:
                    dying.setdefault(last[k], []).append((base + offset[k], size[k]))
            lib = self.lib
            for i, regions in dying.items():
                def poisoned(st, op=self.ops_main[i], regions=tuple(regions)):
                    op(st)
                    for ptr, nbytes in regions:
                        cabi.check(lib.ld_memset_bytes(ptr, 0xFF, nbytes, st), "pool verify")
                self.ops_main[i] = poisoned
            self.pool_stats["verify_poisoned_buffers"] = sum(len(r) for r in dying.values())

    def slot(self):
        s = self._slot_cursor
        self._slot_cursor += 1
        assert s < self.stats.shape[0]
        return self.stats[s]

    def t_ptr(self):
        """The step-counter argument of the launches that apply FiLM: none -- in table mode the row of the current step is
        at a fixed address (film_cur), with per-sample timesteps the rows are indexed by the batch element."""
        return None

    def src(self, t, c, stride=0, ups=0, gn=None, act=cabi.ACT_NONE, film=None):
        s = cabi.Src()
        s.data, s.C, s.pix_stride, s.upsample = t.data_ptr(), c, stride, ups
        if gn is not None:
            stats, gamma, beta, groups = gn
            s.gn_stats, s.gn_gamma, s.gn_beta, s.gn_groups = stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), groups
        s.act = act
        if film is not None and self.table_T:          # the step's row, gathered by ld_step_begin_film
            s.film = self.film_cur.data_ptr() + 4 * self._film_off[id(film)]
            s.film_tstride = s.film_bstride = 0
        elif film is not None:                          # per-sample timesteps (Unet.forward): a row per batch element
            s.film = film.data_ptr()
            s.film_tstride = 0
            s.film_bstride = 2 * c
        self.keep.append(t)
        s._tensor = t                                  # (python-side only: the builders declare it as read; lost when the struct is copied)
        return s

    def _call(self, ops, fn, args, what, meta=None, reads=(), writes=()):
        """``reads`` / ``writes``: the tensors this launch reads and writes (the pool's liveness comes from these declarations;
        the reflection scan of ``args`` cross-checks them and patches the pointers: _pool_buffers)."""
        self.keep.append(args)
        ref = C.byref(args)
        ops.append(lambda st, fn=fn, ref=ref, what=what: cabi.check(fn(ref, st), what))
        if ops is self.ops_main:
            self.meta[len(ops) - 1] = dict(what=what, **(meta or {}))
            self.decl[len(ops) - 1] = ([t for t in reads if t is not None], [t for t in writes if t is not None])

    def _raw(self, ops, fn, what, nbytes=0, flops=0, reads=(), writes=()):
        ops.append(fn)
        if ops is self.ops_main:
            self.meta[len(ops) - 1] = dict(what=what, family=what, bytes=nbytes, flops=flops)
            self.decl[len(ops) - 1] = ([t for t in reads if t is not None], [t for t in writes if t is not None])

    def conv3(self, ops, srcs, wname, cout, h, w, stats=None, groups=8, weight=None, bias=None, addend=None, side=None):
        """``side`` = (packed 1x1 weight, bias): the ResnetBlock's res_conv over the same input as a second output of the launch
        (ld_conv3x3_args.side_*); returns (out, side_out) then."""
        a = cabi.Conv3x3Args()
        for i, s in enumerate(srcs):
            a.src[i] = s
        a.nsrc = len(srcs)
        weight = self.P["w"][wname + ".weight"] if weight is None else weight
        bias = self.f32[wname + ".bias"] if bias is None else bias
        a.weight, a.bias = weight.data_ptr(), bias.data_ptr()
        a.weight_terms = 2 if weight.data_ptr() in self.P["terms2"] else 1
        a.addend = cabi.ptr(addend)
        self.keep += [weight, bias, addend]
        out = self.buf(h, w, cout)
        a.out = out.data_ptr()
        side_out = None
        if side is not None:
            side_out = self.buf(h, w, cout)
            a.side_weight, a.side_bias, a.side_out = side[0].data_ptr(), side[1].data_ptr(), side_out.data_ptr()
            self.keep += [side[0], side[1]]
        if stats is not None:
            a.out_stats, a.out_groups = stats.data_ptr(), groups
        a.B, a.H, a.W, a.Cout = self.B, h, w, cout
        a.t_ptr = self.t_ptr()
        a.dtype = self.dt
        cin = sum(s.C for s in srcs)
        npx = self.B * h * w
        mt4 = cout % 64 == 0 and ((w + 15) // 16) * ((h + 7) // 8) * (cout // 64) * self.B >= 256
        blocks16 = ((w + 15) // 16) * ((h + 15) // 16) * (cout // (64 if mt4 else 32)) * self.B
        big = blocks16 >= 512 and h >= 16 and not mt4
        big = big or (mt4 and h >= 16 and blocks16 >= self._kt("conv_big4_min"))    # mirrors conv3x3.hip:dispatch
        dname = {cabi.LD_F32: "f32", cabi.LD_BF16: "bf16", cabi.LD_F16: "f16"}[self.dt]
        if side is not None:
            big = False                                # mirrors conv3x3.hip:dispatch (the side output rides on the 8-row tiles)
        fam = f"conv3x3<{dname},{4 if mt4 else 2},{4 if big else 2}>"
        ck = 16 if self.dt == cabi.LD_F32 else 32
        if side is not None:
            pass                                       # (only the generic kernel has the side output)
        elif (cout == 32 and len(srcs) == 1 and cin == ck and addend is None and h >= 32 and w >= 32 and h % 16 == 0
                and w % 16 == 0 and (w // 16) * (h // 16) * self.B >= self._kt("conv_c32_min_tiles")
                and self._kt("conv_c32") and a.weight_terms != 2):
            fam = f"conv3x3_c32<{dname}>"              # the persistent LDS-DMA kernel takes it (conv3x3_c32.hip)
        elif (self.dt != cabi.LD_F32 and cout == 32 and addend is None and a.weight_terms != 2 and h % 16 == 0 and w % 16 == 0
              and (w // 16) * (h // 16) * self.B >= self._kt("conv_s32_min_tiles") and not any(s.upsample for s in srcs)
              and cin in (32, 64) and (len(srcs) == 1 or all(s.C == 32 for s in srcs)) and groups in (1, 2, 4, 8)):
            pro = any(s.gn_stats for s in srcs)        # mirrors ld_conv3x3_s32_try (conv3x3_s32.hip)
            bit = 2 if pro else (1 if cin == 32 else 4)
            if (self._kt("conv_s32") & bit) and (not pro or (len(srcs) == 1 and srcs[0].C == 32 and srcs[0].gn_groups == 8)):
                fam = f"conv3x3_s32<{dname}>"
        in_el = sum(s.C * (npx // 4 if s.upsample else npx) for s in srcs)
        # (algorithmic bytes / flops, SURVEY 8d: with a side output the launch also carries res_conv's -- its input, read here once
        #  for both, counted for both as the reference's two modules would)
        side_b = (in_el + npx * cout + cin * cout) * self.esize + 4 * cout if side is not None else 0
        self._call(ops, self.lib.ld_conv3x3, a, "conv3x3 " + wname + (" + res_conv" if side is not None else ""),
                   dict(family=fam, bytes=(in_el + npx * cout + 9 * cin * cout) * self.esize + 4 * cout + side_b,
                        flops=2 * (9 + (1 if side is not None else 0)) * cin * cout * npx, shape=f"{cin}->{cout}@{h}x{w}"),
                   reads=[s._tensor for s in srcs] + [addend], writes=[out, side_out])
        return out if side is None else (out, side_out)

    def conv1(self, ops, srcs, weight, cout, h, w, bias=None, epi=cabi.EPI_PLAIN, unshuffle=0, rms_in=0,
              bstride=0, g2=None, residual=None, what="conv1x1", out=None, kmax_out=None, gn_tail=None):
        a = cabi.Conv1x1Args()
        for i, s in enumerate(srcs):
            a.src[i] = s
        a.nsrc, a.unshuffle, a.rms_in = len(srcs), unshuffle, rms_in
        a.weight, a.weight_bstride = weight.data_ptr(), bstride
        a.weight_terms = 2 if weight.data_ptr() in self.P["terms2"] else 1
        a.bias = cabi.ptr(bias)
        a.epilogue, a.hidden, a.q_scale = epi, self.cfg.hidden, self.cfg.attn_dim_head ** -0.5
        a.g2, a.residual = cabi.ptr(g2), cabi.ptr(residual)
        a.kmax_out = cabi.ptr(kmax_out)
        if gn_tail is not None:
            a.gn_tail = gn_tail
        out = self.buf(h, w, cout) if out is None else out
        a.out = out.data_ptr()
        a.B, a.H, a.W, a.Cout, a.dtype = self.B, h, w, cout, self.dt
        self.keep += [weight, bias, g2, residual]
        cin = sum(s.C for s in srcs) * (4 if unshuffle else 1)
        npx = self.B * h * w
        el = npx * cin + npx * cout + cin * cout * (self.B if bstride else 1) + (npx * cout if (residual is not None or gn_tail is not None) else 0)
        self._call(ops, self.lib.ld_conv1x1, a, what,
                   dict(family="conv1x1", bytes=el * self.esize, flops=2 * cin * cout * npx, shape=f"{cin}->{cout}@{h}x{w}"),
                   reads=[s._tensor for s in srcs] + [residual, gn_tail._tensor if gn_tail is not None else None], writes=[out])
        return out

    def gn_apply(self, ops, a_src, b_src, h, w, c, final_act=cabi.ACT_NONE, pool=0):
        g = cabi.GnApplyArgs()
        g.a = a_src
        if b_src is not None:
            g.b = b_src
        g.final_act, g.pool = final_act, pool
        out = self.buf(h // 2 if pool else h, w // 2 if pool else w, c)
        g.out = out.data_ptr()
        g.B, g.H, g.W, g.t_ptr, g.dtype = self.B, h, w, self.t_ptr(), self.dt
        nel = self.B * h * w * c
        self._call(ops, self.lib.ld_gn_apply, g, "gn_apply",
                   dict(family="gn_apply", bytes=(nel * (2 if b_src is not None else 1) + nel // (4 if pool else 1)) * self.esize, flops=0),
                   reads=[a_src._tensor, b_src._tensor if b_src is not None else None], writes=[out])
        return out

    # ------------------------------------------------------------------ time embedding / FiLM
    def _build_time(self):
        cfg, f, lib = self.cfg, self.f32, self.lib
        if self.table_T:
            times = torch.arange(self.table_T, dtype=torch.int32, device=self.dev)
        else:
            times = self.times
        self.keep.append(times)
        freqs = self.P["freqs"]
        n, td = self.rows, cfg.time_dim
        if cfg.learned_sinusoidal_dim:            # RandomOrLearnedSinusoidalPosEmb (ddpm.py:151-165): [t, sin(2 pi w t), cos(2 pi w t)]
            wts = f["time_mlp.0.weights"]
            self.ops_time.append(lambda st: cabi.check(lib.ld_time_mlp_fourier(
                times.data_ptr(), n, wts.data_ptr(), cfg.learned_sinusoidal_dim, f["time_mlp.1.weight"].data_ptr(),
                f["time_mlp.1.bias"].data_ptr(), f["time_mlp.3.weight"].data_ptr(), f["time_mlp.3.bias"].data_ptr(),
                td, self.temb.data_ptr(), st), "time_mlp (learned Fourier features)"))
        else:
            self.ops_time.append(lambda st: cabi.check(lib.ld_time_mlp(
                times.data_ptr(), n, freqs.data_ptr(), cfg.dim, f["time_mlp.1.weight"].data_ptr(),
                f["time_mlp.1.bias"].data_ptr(), f["time_mlp.3.weight"].data_ptr(), f["time_mlp.3.bias"].data_ptr(),
                td, self.temb.data_ptr(), st), "time_mlp"))
        for name in f:
            if name.endswith(".mlp.1.weight") and not name.startswith("conv_fusion"):
                p = name[:-len(".mlp.1.weight")]
                two_c = f[name].shape[0]
                film = torch.zeros(n, two_c, dtype=torch.float32, device=self.dev)
                self.films[p] = film
                self._film_off[id(film)] = self.film_row
                self.film_row += two_c
                w, b = f[name], f[p + ".mlp.1.bias"]
                self.ops_time.append(lambda st, w=w, b=b, film=film, two_c=two_c: cabi.check(lib.ld_film(
                    self.temb.data_ptr(), n, td, w.data_ptr(), b.data_ptr(), two_c, film.data_ptr(), st), "film"))

    # ------------------------------------------------------------------ blocks
    def resnet_block(self, ops, p, srcs_fn, cin_total, cout, h, w, res_tensor=None):
        """ddpm.py:188-212.  ``srcs_fn()`` returns fresh ld_src structs of the block input (one or two
        concatenated tensors); ``res_tensor`` is the single input tensor when cin == cout."""
        f, G = self.f32, self.cfg.resnet_block_groups
        s1, s2 = self.slot(), self.slot()
        # res_conv in block1's launch (round 6, 16-bit storage, one-term weights, maps up to side_res_conv_max_px): the tail below
        # is then a gn_apply with res_conv(x) as its second operand instead of a conv1x1 that walks the block's input a second time
        wres = self.P["w"].get(p + ".res_conv.weight")
        side = None
        px_ok = (h * w in self.tn.side_res_conv_px) if self.tn.side_res_conv_px is not None else h * w <= self.tn.side_res_conv_max_px
        # (not in the accuracy mode: res_conv's output is rounded to the storage type on its way to the tail, one rounding the fused
        #  tail does not have -- with two-term weights that rounding shows in the chain's max-abs distance, golden G16)
        if (wres is not None and self.tn.side_res_conv and self.dt != cabi.LD_F32 and px_ok and self.net.weight_split_levels == 0
                and wres.data_ptr() not in self.P["terms2"] and self.P["w"][p + ".block1.proj.weight"].data_ptr() not in self.P["terms2"]):
            side = (wres, f[p + ".res_conv.bias"])
        rc = None
        if side is not None:
            raw1, rc = self.conv3(ops, srcs_fn(), p + ".block1.proj", cout, h, w, stats=s1, groups=G, side=side)
        else:
            raw1 = self.conv3(ops, srcs_fn(), p + ".block1.proj", cout, h, w, stats=s1, groups=G)
        film = self.films.get(p)
        n1 = self.src(raw1, cout, gn=(s1, f[p + ".block1.norm.weight"], f[p + ".block1.norm.bias"], G),
                      act=cabi.ACT_SILU, film=film)
        if self._separate_act(h * w, cout):
            # small, wide maps: the normalise+FiLM+SiLU prologue would be repeated by every cout-tile workgroup
            # (4x at 256 channels) on the critical path of one-workgroup-per-CU launches; a separate pass over
            # the (L2-resident) tensor measured cheaper (256->256@32^2: 45 -> 25 us + 9 us).  Everything else keeps
            # it fused (saves a full HBM round trip of the activation).
            act1 = self.gn_apply(ops, n1, None, h, w, cout)
            n1 = self.src(act1, cout)
        raw2 = self.conv3(ops, [n1], p + ".block2.proj", cout, h, w, stats=s2, groups=G)
        n2 = self.src(raw2, cout, gn=(s2, f[p + ".block2.norm.weight"], f[p + ".block2.norm.bias"], G),
                      act=cabi.ACT_SILU)
        if rc is not None:
            out = self.gn_apply(ops, n2, self.src(rc, cout), h, w, cout)          # out = SiLU(GN(raw2)) + res_conv(x)
        elif (p + ".res_conv.weight") in f:
            # block tail fused into the res_conv epilogue: out = res_conv(x) + SiLU(GN(raw2))
            out = self.conv1(ops, srcs_fn(), self.P["w"][p + ".res_conv.weight"], cout, h, w,
                             bias=f[p + ".res_conv.bias"], epi=cabi.EPI_GN_TAIL, gn_tail=n2, what="res_conv+tail " + p)
        else:
            assert res_tensor is not None and cin_total == cout
            out = self.gn_apply(ops, n2, self.src(res_tensor, cout), h, w, cout)
        self.named[p] = out
        return out

    def fusion_block_folded(self, ops, x, c, h, w):
        """conv_fusion (a ResnetBlock over cat(trunk, conditioning features), ddpm.py:434-436) with the conditioning
        halves of block1.proj and res_conv precomputed by run_cond: per step only the trunk halves are convolved,
        the constants enter as a pre-statistics addend / an extra residual."""
        f, G, p = self.f32, self.cfg.resnet_block_groups, "conv_fusion"
        p1, p2, zero = self.fusion_const
        s1, s2 = self.slot(), self.slot()
        raw1 = self.conv3(ops, [self.src(x, c)], p + ".block1.proj", c, h, w, stats=s1, groups=G,
                          weight=self.P["w"][p + ".block1.proj.weight#x"], bias=zero, addend=p1)
        n1 = self.src(raw1, c, gn=(s1, f[p + ".block1.norm.weight"], f[p + ".block1.norm.bias"], G),
                      act=cabi.ACT_SILU, film=self.films.get(p))
        if self._separate_act(h * w, c):
            n1 = self.src(self.gn_apply(ops, n1, None, h, w, c), c)
        raw2 = self.conv3(ops, [n1], p + ".block2.proj", c, h, w, stats=s2, groups=G)
        n2 = self.src(raw2, c, gn=(s2, f[p + ".block2.norm.weight"], f[p + ".block2.norm.bias"], G), act=cabi.ACT_SILU)
        out = self.conv1(ops, [self.src(x, c)], self.P["w"][p + ".res_conv.weight#x"], c, h, w, bias=None,
                         epi=cabi.EPI_GN_TAIL, gn_tail=n2, residual=p2, what="res_conv+tail " + p)
        self.named[p] = out
        return out

    def linear_attention_fused(self, ops, p, x, c, h, w):
        """bf16 / fp16: q/k/v never materialised (csrc/linattn_fused.hip)."""
        f, cfg, lib = self.f32, self.cfg, self.lib
        n, hid, B, heads, dt, es = h * w, cfg.hidden, self.B, cfg.attn_heads, self.dt, self.esize
        # one workgroup per (batch, chunk) with a wave per head: 512-pixel chunks give >= 1024 workgroups at 256^2
        # pixels per kvctx workgroup (measured, cfg3: 512 best up to 128^2; 1024 halves the partials at 256^2)
        # (round 2, per launch in the two-sub-batch regime: 64^2 maps with 512-pixel chunks were 32 workgroups walking
        # 16 dependent 32-pixel groups each -- 128-pixel chunks there, 256 at 128^2: kvctx 21.7 -> ~15 us per launch,
        # the reduce + fold launch pays 2 us of it back for the extra partials; tools/exp_kvctx.sh)
        # (end of round 2: 512 instead of 1024 at 256^2 -- 128 partials per image, two workgroups per CU at 4 patches:
        # step -1.0 % on the same box, three alternating rounds)
        # ... at up to 8 patches per launch; 64 patches per GPU measured 0.8 % slower with it and keep 1024)
        rule = self.tn.chunk_rule(B)                      # n >= 65536, n >= 16384, smaller
        chunk_px = rule[0] if n >= 65536 else (rule[1] if n >= 16384 else rule[2])
        nchunks = max(1, min(128, n // chunk_px)) if heads == 4 else max(1, min(32, n // 256))
        # one or two large images per launch (cfg5: 512^2 maps at B = 1): 128 chunks of 2,048 pixels are 128 workgroups on
        # 256 CUs.  Chunks shrink (down to one 256-pixel tile) until the launch has two workgroups per CU; ctxfold
        # combines up to 512 partials.  cfg3 / cfg4 (B >= 4 at 256^2) are untouched: 512 workgroups already.
        if heads == 4 and self.tn.linattn_chunk_px is None:
            while B * nchunks < 512 and nchunks < 512 and n // (2 * nchunks) >= 256:
                nchunks *= 2
        ctx = self._tracked(torch.empty(int(lib.ld_linattn_ctx_part_floats(B, heads, 32, nchunks)), dtype=torch.float32, device=self.dev))
        wfold = self._tracked(torch.empty(B, c * hid, dtype=self.tdt, device=self.dev))
        wout = f[p + ".to_out.0.weight"].reshape(c, hid).contiguous()
        wq, wkv = self.P["wq"][p], self.P["wkv"][p]
        kshift = self.P["kshift"].get(p) if heads == 4 else None
        ksp = kshift.data_ptr() if kshift is not None else None
        qshift = self.P["qshift"].get(p) if heads == 4 else None
        qsp = qshift.data_ptr() if qshift is not None else None
        bias, g2 = f[p + ".to_out.0.bias"], self.P["g2"][p + ".to_out.1.g"]
        out = self.buf(h, w, c)
        self.keep += [ctx, wfold, wout, wq, wkv, kshift, qshift, bias, g2, out, x]
        npx = B * n
        terms = self.P.get("attn_terms", {}).get(p, 1)
        self._raw(ops, lambda st: cabi.check(lib.ld_linattn_kvctx_terms(x.data_ptr(), wkv.data_ptr(), ksp, ctx.data_ptr(), B, n, c,
                                                                        heads, 32, nchunks, dt, terms, st), "linattn_kvctx"),
                  "linattn_kvctx", nbytes=npx * c * es, flops=2 * npx * c * 2 * hid * (1 if kshift is not None else 2) + 2 * npx * hid * 32,
                  reads=[x], writes=[ctx])
        self._raw(ops, lambda st: cabi.check(lib.ld_linattn_ctxfold(ctx.data_ptr(), nchunks, wout.data_ptr(), wfold.data_ptr(), B, c,
                                                                    heads, 32, 1, dt, st), "linattn_ctxfold"),
                  "linattn_ctxfold", nbytes=B * c * hid * es, flops=2 * B * c * hid * 32, reads=[ctx], writes=[wfold])
        scale = cfg.attn_dim_head ** -0.5
        self._raw(ops, lambda st: cabi.check(lib.ld_linattn_out_terms(x.data_ptr(), wq.data_ptr(), qsp, wfold.data_ptr(), bias.data_ptr(),
                                                                      g2.data_ptr(), out.data_ptr(), B, n, c, scale, dt, terms, st),
                                             "linattn_out"),
                  "linattn_out", nbytes=2 * npx * c * es, flops=2 * npx * hid * c * 2, reads=[x, wfold], writes=[out])
        return out

    def linear_attention(self, ops, p, x, c, h, w):
        """ddpm.py:214-251 (+ residual of the caller, :425/:444)."""
        if self.dt != cabi.LD_F32 and c in (32, 64, 128) and p in self.P["wq"]:
            return self.linear_attention_fused(ops, p, x, c, h, w)
        f, cfg, lib = self.f32, self.cfg, self.lib
        n, hid, B = h * w, cfg.hidden, self.B
        kmax = self.kmax_arena[self._kmax_cursor]
        self._kmax_cursor += 1
        qkv = self.conv1(ops, [self.src(x, c)], self.P["w"][p + ".to_qkv.weight"], 3 * hid, h, w,
                         epi=cabi.EPI_QKV_LINEAR, rms_in=1, what="to_qkv " + p, kmax_out=kmax)
        nchunks = max(1, min(32, n // 256))       # >= 64 pixels per wave, enough workgroups at small n
        ctx = torch.empty(int(lib.ld_linattn_ctx_part_floats(B, cfg.attn_heads, 32, nchunks)),
                          dtype=torch.float32, device=self.dev)
        wfold = torch.empty(B, c * hid, dtype=self.tdt, device=self.dev)
        wout = f[p + ".to_out.0.weight"].reshape(c, hid).contiguous()
        self.keep += [kmax, ctx, wfold, wout, qkv]
        dt, heads = self.dt, cfg.attn_heads
        es = self.esize
        self._raw(ops, lambda st: cabi.check(lib.ld_linattn_ctx(qkv.data_ptr(), kmax.data_ptr(), ctx.data_ptr(),
                                                                B, n, heads, 32, nchunks, dt, st), "linattn_ctx"),
                  "linattn_ctx", nbytes=2 * B * n * hid * es, flops=2 * B * n * hid * 32, reads=[qkv], writes=[ctx])
        self._raw(ops, lambda st: cabi.check(lib.ld_linattn_ctxfold(ctx.data_ptr(), nchunks, wout.data_ptr(), wfold.data_ptr(),
                                                                    B, c, heads, 32, 0, dt, st), "linattn_ctxfold"),
                  "linattn_ctxfold", nbytes=B * c * hid * es, flops=2 * B * c * hid * 32, reads=[ctx], writes=[wfold])
        q = self.src(qkv, hid, stride=3 * hid)
        return self.conv1(ops, [q], wfold, c, h, w, bias=f[p + ".to_out.0.bias"], epi=cabi.EPI_RMS_RES,
                          bstride=c * hid * self.esize, g2=self.P["g2"][p + ".to_out.1.g"], residual=x,
                          what="lin to_out " + p)

    def full_attention(self, ops, p, x, c, h, w):
        """ddpm.py:253-282 + attend.py:84-113 (+ residual of the caller)."""
        f, cfg, lib = self.f32, self.cfg, self.lib
        n, hid, B = h * w, cfg.hidden, self.B
        qkv = self.conv1(ops, [self.src(x, c)], self.P["w"][p + ".to_qkv.weight"], 3 * hid, h, w,
                         epi=cabi.EPI_QKV_FULL, rms_in=1, what="to_qkv " + p)
        att = self.buf(h, w, hid)
        dt, heads = self.dt, cfg.attn_heads
        self.keep += [qkv, att]
        self._raw(ops, lambda st: cabi.check(lib.ld_attention(qkv.data_ptr(), att.data_ptr(), B, n, heads, 32, dt, st),
                                             "attention"),
                  "attention", nbytes=4 * B * n * hid * self.esize, flops=4 * B * n * n * hid, reads=[qkv], writes=[att])
        return self.conv1(ops, [self.src(att, hid)], self.P["w"][p + ".to_out.weight"], c, h, w,
                          bias=f[p + ".to_out.bias"], epi=cabi.EPI_RES, residual=x, what="full to_out " + p)

    def attention(self, ops, p, x, c, h, w, full):
        out = (self.full_attention if full else self.linear_attention)(ops, p, x, c, h, w)
        self.named[p] = out
        return out

    # ------------------------------------------------------------------ conditioning encoder
    def _basic_block(self, ops, p, x, cin, cmid, cout, h, w, pool, image=False):
        """unet_model.py:8-51 (+ MaxPool2d of ResUnet.forward when pool)."""
        f, lib, G = self.f32, self.lib, 16
        s1, s2, s3 = self.slot(), self.slot(), self.slot()

        def image_conv(wname, c_out, stats):
            out = self.buf(h, w, c_out)
            wt, bs = f[wname + ".weight"], f[wname + ".bias"]
            B, dt = self.B, self.dt
            self.keep += [out, wt, bs]
            ops.append(lambda st: cabi.check(lib.ld_conv_image(
                self.cond_in.data_ptr(), wt.data_ptr(), bs.data_ptr(), out.data_ptr(), stats.data_ptr(), G,
                B, cin, h, w, 3, dt, st), "conv_image " + wname))
            return out

        if image:
            a1 = image_conv(p + ".convblock.0", cmid, s1)
            idr = image_conv(p + ".identity.0", cout, s3)
        else:
            a1 = self.conv3(ops, [self.src(x, cin)], p + ".convblock.0", cmid, h, w, stats=s1, groups=G)
            idr = self.conv3(ops, [self.src(x, cin)], p + ".identity.0", cout, h, w, stats=s3, groups=G)
        n1 = self.src(a1, cmid, gn=(s1, f[p + ".convblock.1.weight"], f[p + ".convblock.1.bias"], G),
                      act=cabi.ACT_RELU)
        a2 = self.conv3(ops, [n1], p + ".convblock.3", cout, h, w, stats=s2, groups=G)
        na = self.src(a2, cout, gn=(s2, f[p + ".convblock.4.weight"], f[p + ".convblock.4.bias"], G))
        nb = self.src(idr, cout, gn=(s3, f[p + ".identity.1.weight"], f[p + ".identity.1.bias"], G))
        return self.gn_apply(ops, na, nb, h, w, cout, final_act=cabi.ACT_RELU, pool=1 if pool else 0)

    def _build_cond(self):
        cfg, ops, H, W = self.cfg, self.ops_cond, self.H, self.W
        x = self._basic_block(ops, "cond_model.residual_conv1.0", None, cfg.cond_in_channels, 32, 32, H, W,
                              pool=True, image=True)
        x = self._basic_block(ops, "cond_model.residual_conv2.0", x, 32, 32, 64, H // 2, W // 2, pool=True)
        early = cfg.cond_early_exit
        x = self._basic_block(ops, "cond_model.residual_conv3.0", x, 64, 64, 128, H // 4, W // 4, pool=not early)
        if early:
            return x
        return self._basic_block(ops, "cond_model.mid_conv.0", x, 128, 128, 256, H // 8, W // 8, pool=False)

    # ------------------------------------------------------------------ main trunk (ddpm.py:413-451)
    def _build_main(self):
        cfg, ops, f, lib, B = self.cfg, self.ops_main, self.f32, self.lib, self.B
        H, W = self.H, self.W
        r = self.buf(H, W, cfg.init_dim)
        wi, bi = f["init_conv.weight"], f["init_conv.bias"]
        if self.dt != cabi.LD_F32 and cfg.init_dim == 32:        # implicit GEMM on MFMA (hi/lo split: fp32-accurate)
            wstem = self.P["stem"]
            self.keep.append(wstem)

            def stem_begin(st, delta, idx_ptr, t_table):        # init_conv + the head-of-step work in ONE launch (ld_conv_stem_begin)
                za, ba, zb, bb = self._begin_args
                rows, row_floats, cur = self._film_args
                g = cabi.StepBeginArgs(za, ba, zb, bb, self._t_dev_ptr, int(delta), idx_ptr, t_table, rows, row_floats, cur)
                cabi.check(lib.ld_conv_stem_begin(self.x_in.data_ptr(), wstem.data_ptr(), bi.data_ptr(), r.data_ptr(), B, cfg.channels,
                                                  H, W, self.dt, C.byref(g), st), "init_conv + step_begin")
            self._stem_begin = stem_begin
            self._raw(ops, lambda st: cabi.check(lib.ld_conv_stem(
                self.x_in.data_ptr(), wstem.data_ptr(), bi.data_ptr(), r.data_ptr(), B, cfg.channels, H, W, self.dt, st),
                "init_conv"), "conv_image7x7",
                nbytes=B * H * W * (4 * cfg.channels + self.esize * cfg.init_dim), flops=2 * 49 * cfg.channels * cfg.init_dim * B * H * W, writes=[r])
        else:
            self._raw(ops, lambda st: cabi.check(lib.ld_conv_image(
                self.x_in.data_ptr(), wi.data_ptr(), bi.data_ptr(), r.data_ptr(), None, 1, B, cfg.channels, H, W, 7,
                self.dt, st), "init_conv"), "conv_image7x7",
                nbytes=B * H * W * (4 * cfg.channels + self.esize * cfg.init_dim), flops=2 * 49 * cfg.channels * cfg.init_dim * B * H * W, writes=[r])
        self.named["init_conv"] = r
        x, c, h, w = r, cfg.init_dim, H, W
        skips = []
        io, n = cfg.in_out, len(cfg.in_out)
        for i, (cin, cout) in enumerate(io):
            p = f"downs.{i}"
            x = self.resnet_block(ops, p + ".0", lambda x=x, c=c: [self.src(x, c)], c, c, h, w, res_tensor=x)
            skips.append((x, c))
            x = self.resnet_block(ops, p + ".1", lambda x=x, c=c: [self.src(x, c)], c, c, h, w, res_tensor=x)
            x = self.attention(ops, p + ".2", x, c, h, w, cfg.full_attn[i])
            skips.append((x, c))
            if i < n - 1:
                h, w = h // 2, w // 2
                x = self.conv1(ops, [self.src(x, c)], self.P["w"][p + ".3.1.weight"], cout, h, w,
                               bias=f[p + ".3.1.bias"], unshuffle=1, what="downsample " + p)
            else:
                x = self.conv3(ops, [self.src(x, c)], p + ".3", cout, h, w)
            self.named[p + ".3"] = x
            c = cout
        x = self.resnet_block(ops, "mid_block1", lambda x=x, c=c: [self.src(x, c)], c, c, h, w, res_tensor=x)
        x = self.attention(ops, "mid_attn", x, c, h, w, True)
        x = self.resnet_block(ops, "mid_block2", lambda x=x, c=c: [self.src(x, c)], c, c, h, w, res_tensor=x)
        feat = self.cond_feat
        if self.fusion_const is not None:
            x = self.fusion_block_folded(ops, x, c, h, w)
        else:
            x = self.resnet_block(ops, "conv_fusion", lambda x=x, c=c: [self.src(x, c), self.src(feat, c)], 2 * c, c, h, w)
        for j, ((cin, cout), full) in enumerate(zip(reversed(io), reversed(cfg.full_attn))):
            p = f"ups.{j}"
            for k in (0, 1):
                sk, sc = skips.pop()
                x = self.resnet_block(ops, f"{p}.{k}", lambda x=x, c=c, sk=sk, sc=sc: [self.src(x, c), self.src(sk, sc)],
                                      c + sc, cout, h, w)
                c = cout
            x = self.attention(ops, p + ".2", x, c, h, w, full)
            if j < n - 1:
                h, w = h * 2, w * 2
                x = self.conv3(ops, [self.src(x, c, ups=1)], p + ".3.1", cin, h, w)
            else:
                x = self.conv3(ops, [self.src(x, c)], p + ".3", cin, h, w)
            self.named[p + ".3"] = x
            c = cin
        if self.tn.recompute_stem and self._track is not None and self.dt != cabi.LD_F32 and cfg.init_dim == 32:
            # the final block concatenates init_conv's output, which would otherwise stay live for the whole evaluation
            # (one of six 16 MB tensors at the peak of a 4-patch plan): evaluate init_conv again here instead
            r2 = self.buf(H, W, cfg.init_dim)
            wstem2 = self.P["stem"]
            self._raw(ops, lambda st, r2=r2: cabi.check(lib.ld_conv_stem(
                self.x_in.data_ptr(), wstem2.data_ptr(), bi.data_ptr(), r2.data_ptr(), B, cfg.channels, H, W, self.dt, st),
                "init_conv (again)"), "conv_image7x7",
                nbytes=B * H * W * (4 * cfg.channels + self.esize * cfg.init_dim), flops=2 * 49 * cfg.channels * cfg.init_dim * B * H * W, writes=[r2])
            r_cat = r2                                   # (``r`` itself stays bound: the first launch's closures write it)
        else:
            r_cat = r
        x = self.resnet_block(ops, "final_res_block", lambda x=x, c=c, r_cat=r_cat: [self.src(x, c), self.src(r_cat, cfg.init_dim)],
                              c + cfg.init_dim, cfg.dim, h, w)
        wf = f["final_conv.weight"].reshape(cfg.out_dim, cfg.dim).contiguous()
        bf = f["final_conv.bias"]
        self.keep += [wf, bf, x, r]
        self.final = (x, wf, bf)              # operands of the last op (diffusion.py fuses it with the sampler update)
        self.live_out.append(x)               # ... which reads x AFTER the launch list: declared live to the end of the evaluation
        self._raw(ops, lambda st, x=x: cabi.check(lib.ld_final_conv(
            x.data_ptr(), wf.data_ptr(), bf.data_ptr(), self.model_out.data_ptr(), B, H, W, cfg.dim, cfg.out_dim,
            self.dt, st), "final_conv"), "final_conv",
            nbytes=B * H * W * (self.esize * cfg.dim + 4 * cfg.out_dim), flops=2 * cfg.dim * cfg.out_dim * B * H * W, reads=[x])

    def __del__(self):
        try:
            if getattr(self, "_cond_graph", None) is not None:
                self.lib.ld_graph_destroy(self._cond_graph)
        except Exception:
            pass

    # ------------------------------------------------------------------ execution
    def run_time(self, st):
        for op in self.ops_time:
            op(st)

    def run_cond(self, st):
        self.cond_version = getattr(self, "cond_version", 0) + 1
        s = self.stats[:self.cond_slots]
        self.lib.ld_range_push(b"encoder")
        try:
            cabi.check(self.lib.ld_memset_zero(s.data_ptr(), s.numel() * 8, st), "memset")
            for op in self.ops_cond:
                op(st)
        finally:
            self.lib.ld_range_pop()

    def run_cond_replayed(self, stream):
        """``run_cond`` as ONE replayed HIP graph on ``stream`` (a torch stream that is current): the encoder of a sample
        is ~45 dependent launches (0.4 ms of host-paced issue per sub-batch at the head of every sample, VERDICT r3 item
        5); its buffers and arguments are static, so after the first evaluation -- eager, because lazy
        hipFuncSetAttribute calls must not happen inside a capture -- the launch sequence is captured once and every
        later sample replays it with one graph launch."""
        st = stream.cuda_stream
        g = getattr(self, "_cond_graph", None)
        if g is None:
            self.run_cond(st)
            stream.synchronize()
            s = self.stats[:self.cond_slots]
            cabi.check(self.lib.ld_graph_begin(st), "graph_begin")
            try:
                cabi.check(self.lib.ld_memset_zero(s.data_ptr(), s.numel() * 8, st), "memset")
                for op in self.ops_cond:
                    op(st)                     # recorded, not run
            finally:
                g = C.c_void_p()
                rc = self.lib.ld_graph_end(st, C.byref(g))
            cabi.check(rc, "graph_end")
            self._cond_graph = g
            # the runtime finishes building an executable graph on its FIRST launch (0.3-1 ms of host time, by the box):
            # take it here, with the capture, not at the head of the second sample (the encoder is idempotent)
            if self.tn.graph_prewarm:
                cabi.check(self.lib.ld_graph_launch(g, st), "graph_launch")
            return
        self.cond_version = getattr(self, "cond_version", 0) + 1
        self.lib.ld_range_push(b"encoder (graph replay)")
        try:
            cabi.check(self.lib.ld_graph_launch(g, st), "graph_launch")
        finally:
            self.lib.ld_range_pop()

    def run_main(self, st, skip_final=False, step_delta=0, idx_ptr=None, t_table=None):
        """One denoiser evaluation.  ``skip_final``: stop before final_conv (the caller runs ld_final_step).
        ``step_delta``: added to the device step counter by the evaluation's first launch (ld_step_begin zeroes the
        statistics slots this plan uses, the k-max arena if the unfused linear attention is on the plan, and moves
        the counter -- one launch where two memsets and ld_step_add were three)."""
        self.lib.ld_range_push(b"step")            # roctx: host-side issue of one denoiser evaluation (or its capture)
        try:
            ops = self.ops_main[:-1] if skip_final else self.ops_main
            stem_begin = getattr(self, "_stem_begin", None) if self.tn.fused_step_begin else None
            if stem_begin is not None:             # 16-bit storage: the head-of-step work rides in init_conv's launch
                stem_begin(st, 0 if idx_ptr is not None else int(step_delta), idx_ptr, t_table)
                ops = ops[1:]
            elif idx_ptr is not None:              # strided sampling: t_dev = t_table[++idx] (ld_step_begin)
                cabi.check(self.lib.ld_step_begin_film(*self._begin_args, self._t_dev_ptr, 0, idx_ptr, t_table, *self._film_args, st),
                           "step_begin")
            else:
                cabi.check(self.lib.ld_step_begin_film(*self._begin_args, self._t_dev_ptr, int(step_delta), None, None,
                                                       *self._film_args, st), "step_begin")
            for op in ops:
                op(st)
        finally:
            self.lib.ld_range_pop()

    def run_main_timed(self, st, acc):
        """Like run_main, inside a per-launch timing session of the library (ld_timing_*: every kernel launch carries
        its own start/stop events, so the times are kernel execution times as rocprofv3 reports them); adds each
        op's kernel time to ``acc[index] = [ms_total, launches]`` (bench.py's per-kernel roofline leg)."""
        lib = self.lib
        cabi.check(lib.ld_step_begin_film(*self._begin_args, self._t_dev_ptr, 0, None, None, *self._film_args, st), "step_begin")
        cabi.check(lib.ld_timing_begin(8 * len(self.ops_main)), "timing_begin")
        marks = [0]
        try:
            for op in self.ops_main:
                op(st)
                marks.append(lib.ld_timing_count())
        finally:
            n = max(1, lib.ld_timing_count())
            ms, cnt = (C.c_float * n)(), C.c_int()
            rc = lib.ld_timing_end(ms, n, C.byref(cnt))
        cabi.check(rc, "timing_end")
        for i in range(len(self.ops_main)):
            a = acc.setdefault(i, [0.0, 0])
            a[0] += sum(ms[marks[i]:marks[i + 1]])
            a[1] += 1

    def set_step(self, t):
        self.t_dev.fill_(int(t))
