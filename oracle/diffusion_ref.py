"""ORACLE (test infrastructure, never shipped on the product path).

CPU restatement of the sampling half of the reference's ``GaussianDiffusion``
(/root/reference/ddpm.py:455-1125): schedule buffers, x0/eps conversions, the OOD/IND branch
logic of ``model_predictions`` (:668-766), the fusion step of ``p_mean_variance`` (:768-838),
``p_sample`` (:841-860), the DDPM loop (:930-977), the DDIM loop (:980-1075) and the
``sample`` dispatcher (:1078-1125).

Differences in *form* only: the reference mutates ``self.config`` during sampling (a hidden state
machine); here the phase (BRANCH -> JOINT) is an explicit local variable and
options are an immutable ``SamplerOptions``.  Host/device ping-pong, ``np.save`` side effects,
prints and the per-step ``.cpu()`` history lists are not reproduced.  Noise comes from a
``noise(shape) -> tensor`` callable, called in the reference's draw order (x_T first).

Pinned against the real reference by ``tools/make_goldens.py`` (see oracle/unet_ref.py header).
"""
import math
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import unet_ref


# ----------------------------------------------------------------------------- schedules
def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def beta_schedule(name, T):
    """float64 numpy restatement of ddpm.py:460-494."""
    if name == "linear":
        k = 1000.0 / T
        return np.linspace(k * 1e-4, k * 2e-2, T, dtype=np.float64)
    t = np.linspace(0, T, T + 1, dtype=np.float64) / T
    if name == "cosine":
        s = 0.008
        abar = np.cos((t + s) / (1 + s) * math.pi * 0.5) ** 2
    elif name == "sigmoid":
        start, end, tau = -3.0, 3.0, 1.0
        # quirk: ddpm.py:489-490 builds v_start / v_end with torch.tensor(python float) => they are
        # float32 sigmoids, promoted to float64 afterwards.  Reproduced, or betas differ by ~3e-7.
        v0 = float(torch.tensor(start / tau).sigmoid())
        v1 = float(torch.tensor(end / tau).sigmoid())
        abar = (-_sigmoid((t * (end - start) + start) / tau) + v1) / (v1 - v0)
    else:
        raise ValueError(name)
    abar = abar / abar[0]
    return np.clip(1.0 - abar[1:] / abar[:-1], 0, 0.999)


def schedule_buffers(name, T, objective="pred_x0"):
    """ddpm.py:547-615 -> dict of float32 torch tensors (13 buffers)."""
    b = beta_schedule(name, T)
    a = 1.0 - b
    abar = np.cumprod(a)
    prev = np.concatenate([[1.0], abar[:-1]])
    pv = b * (1.0 - prev) / (1.0 - abar)
    snr = abar / (1 - abar)
    lw = {"pred_noise": snr / snr, "pred_x0": snr, "pred_v": snr / (snr + 1)}[objective]
    d = dict(
        betas=b, alphas_cumprod=abar, alphas_cumprod_prev=prev,
        sqrt_alphas_cumprod=np.sqrt(abar),
        sqrt_one_minus_alphas_cumprod=np.sqrt(1.0 - abar),
        log_one_minus_alphas_cumprod=np.log(1.0 - abar),
        sqrt_recip_alphas_cumprod=np.sqrt(1.0 / abar),
        sqrt_recipm1_alphas_cumprod=np.sqrt(1.0 / abar - 1),
        posterior_variance=pv,
        posterior_log_variance_clipped=np.log(np.maximum(pv, 1e-20)),
        posterior_mean_coef1=b * np.sqrt(prev) / (1.0 - abar),
        posterior_mean_coef2=(1.0 - prev) * np.sqrt(a) / (1.0 - abar),
        loss_weight=lw,
    )
    return {k: torch.from_numpy(v.astype(np.float32)) for k, v in d.items()}


# ----------------------------------------------------------------------------- options
@dataclass(frozen=True)
class SamplerOptions:
    """The config.yaml keys the sampling path reads + ctor arguments."""
    timesteps: int = 1000
    sampling_timesteps: Optional[int] = None
    objective: str = "pred_x0"
    beta_schedule: str = "sigmoid"
    ddim_sampling_eta: float = 0.0
    branch_out: bool = False
    start_intermediate: bool = False
    start_timestep: int = 2
    data: str = "mri"
    mask_x: bool = False
    ood_AD: bool = False
    ood_confidence: bool = False
    classifier: bool = False          # config['classifier']: gate the fused sample with a classifier (ddpm.py:883-916)
    use_gt: bool = False              # config['use_gt']: start from q_sample(hr, use_gt_timestep) (ddpm.py:937-944)
    use_gt_timestep: int = 100
    seed: int = 10


_REPLACE_OUT = ("mnist", "mvtec", "oct", "imagenet")


class RefSampler:
    def __init__(self, model_fn, opts: SamplerOptions, channels, image_size):
        """``model_fn(x, cond, t_long[B]) -> [B,C,H,W]`` is one denoiser evaluation."""
        self.f = model_fn
        self.o = opts
        self.T = opts.timesteps
        self.S = opts.sampling_timesteps or opts.timesteps
        self.buf = schedule_buffers(opts.beta_schedule, self.T, opts.objective)
        self.channels, self.image_size = channels, image_size
        # classifier gate (ddpm.py:622-625 builds a PatchCore model; here any callable x0 -> (score, _, _)).  The
        # flag lives on the object and is never reset (ddpm.py:522): once a sample was accepted, later calls of
        # sample() on the same object skip the classifier.
        self.classifier = None
        self.classifier_flag = 0
        self.pred_cls = None
        # config['mask_x'] as the reference carries it from one sample() call to the next: cleared by the fusion
        # step (ddpm.py:780-781, 1023-1024) and by the all-ones fallback (:1114), set by a classifier rejection
        # (:907-908), re-armed by sample() only when ood_AD / ood_confidence is on (:1106-1108).
        self.mask_x_state = bool(opts.mask_x)

    # --- pointwise conversions (ddpm.py:631-653) ---
    def _c(self, name, t):
        return self.buf[name][t]

    def x0_from_eps(self, x, t, eps):
        return self._c("sqrt_recip_alphas_cumprod", t) * x - self._c("sqrt_recipm1_alphas_cumprod", t) * eps

    def eps_from_x0(self, x, t, x0):
        return (self._c("sqrt_recip_alphas_cumprod", t) * x - x0) / self._c("sqrt_recipm1_alphas_cumprod", t)

    def x0_from_v(self, x, t, v):
        return self._c("sqrt_alphas_cumprod", t) * x - self._c("sqrt_one_minus_alphas_cumprod", t) * v

    def posterior_mean(self, x0, x, t):
        """ddpm.py:659-666."""
        return self._c("posterior_mean_coef1", t) * x0 + self._c("posterior_mean_coef2", t) * x

    def _tvec(self, b, t):
        return torch.full((b,), t, dtype=torch.long)

    # --- training-side forward (ddpm.py:1147-1214): the loss of one batch, no gradient ---
    def _ext(self, name, t, x):
        """extract(buffer, t, x.shape), ddpm.py:455-458: per-sample scalars broadcast over [B,C,H,W]."""
        return self.buf[name][t].reshape(-1, *([1] * (x.dim() - 1)))

    def q_sample(self, x_start, t, noise):
        """ddpm.py:1147-1154."""
        return self._ext("sqrt_alphas_cumprod", t, x_start) * x_start + self._ext("sqrt_one_minus_alphas_cumprod", t, x_start) * noise

    def predict_v(self, x_start, t, noise):
        """ddpm.py:643-647."""
        return self._ext("sqrt_alphas_cumprod", t, x_start) * noise - self._ext("sqrt_one_minus_alphas_cumprod", t, x_start) * x_start

    def p_losses(self, x_start, cond, t, noise, offset_noise=None, offset_noise_strength=0.0):
        """ddpm.py:1156-1201 (self_condition is never enabled by the reference's callers).  ``t`` int64 [B], ``noise`` like
        x_start, ``offset_noise`` [B,C] (drawn after ``noise`` when the strength is positive, :1165-1167)."""
        if offset_noise_strength > 0.0:
            noise = noise + offset_noise_strength * offset_noise[:, :, None, None]
        x = self.q_sample(x_start, t, noise)
        out = self.f(x, cond, t)
        obj = self.o.objective
        target = noise if obj == "pred_noise" else (x_start if obj == "pred_x0" else self.predict_v(x_start, t, noise))
        loss = ((out - target) ** 2).reshape(out.shape[0], -1).mean(dim=1)            # F.mse_loss(reduction='none') + reduce 'b ... -> b'
        loss = loss * self.buf["loss_weight"][t]
        return loss.mean(), loss

    # --- one model evaluation, single branch (ddpm.py:715-761 non-branch arm) ---
    def predict_single(self, x, cond, t, lohi, clip):
        out = self.f(x, cond, self._tvec(x.shape[0], t))
        if self.o.objective == "pred_x0":
            x0 = out.clamp(lohi[0], lohi[1]) if clip else out
            eps = self.eps_from_x0(x, t, x0)
        elif self.o.objective == "pred_noise":
            eps = out
            x0 = self.x0_from_eps(x, t, eps)
            if clip:
                x0 = x0.clamp(lohi[0], lohi[1])
                eps = self.eps_from_x0(x, t, x0)
        else:
            x0 = self.x0_from_v(x, t, out)
            if clip:
                x0 = x0.clamp(lohi[0], lohi[1])
            eps = self.eps_from_x0(x, t, x0)
        return eps, x0

    # --- two-branch evaluation (ddpm.py:671-710, 739-749) ---
    def branch_conditions(self, cond, mask):
        binary = (mask >= 1.0).float()
        lo = 0.5 if self.o.data == "mnist" else 0.95
        cond_out = (cond * binary).float()
        cond_in = (cond * torch.clip(1.0 - binary, lo, 1.0)).float()
        return binary, cond_out, cond_in

    def predict_branches(self, x_out, x_in, cond, mask, t, lohi, clip, mask_x):
        assert self.o.objective == "pred_x0", "branch mode exists only for pred_x0 (ddpm.py:739-749)"
        binary, cond_out, cond_in = self.branch_conditions(cond, mask)
        tv = self._tvec(x_out.shape[0], t)
        # ddpm.py:704-708: for these datasets the OOD-branch network output is thrown away and
        # replaced by cond_out, so the oracle skips that (dead) evaluation; results are identical.
        replaced = mask_x and any(k in self.o.data for k in _REPLACE_OUT) and "mri" not in self.o.data
        m_in = self.f(x_in, cond_in, tv)
        if replaced:
            assert len(torch.unique(binary)) == 2, "mask should be binary"
            m_out = cond_out.clone()
        else:
            m_out = self.f(x_out, cond_out, tv)
            if mask_x:
                assert len(torch.unique(binary)) == 2, "mask should be binary"
                m_out = m_out * binary
                m_out = torch.where(binary == 0.0, torch.tensor(float(lohi[0])), m_out)
        if clip:
            m_out = m_out.clamp(lohi[0], lohi[1])
            m_in = m_in.clamp(lohi[0], lohi[1])
        return (self.eps_from_x0(x_out, t, m_out), m_out), (self.eps_from_x0(x_in, t, m_in), m_in)

    # --- DDPM (ddpm.py:841-860, 930-977) ---
    def _fuse_step(self, xs, cond, mask, t, lohi, mask_x, noise, sigma):
        """Branch evaluation + fusion of one step (ddpm.py:769-810 + 853-858) -> (x_{t-1}, x0, [x_out*m, x_in*(1-m)])."""
        (_, x0o), (_, x0i) = self.predict_branches(xs[0], xs[1], cond, mask, t, lohi, False, mask_x)
        x0o = x0o.clamp(lohi[0], lohi[1])
        x0i = x0i.clamp(lohi[0], lohi[1])
        m = (mask >= 1.0).float()
        x0 = (x0i * (1.0 - m) + x0o)
        xo, xi = xs[0] * m, xs[1] * (1.0 - m)
        assert bool((xo == 0).any()) and bool((xi == 0).any()), "x_out and x_in should be masked"
        x = torch.where(xo == 0.0, xi, xo)
        x0 = x0.clamp(lohi[0], lohi[1])
        mean = self.posterior_mean(x0, x, t)
        z = noise(x.shape) if t > 0 else 0.0
        return mean + sigma * z, x0, [xo, xi]

    def p_sample_loop(self, cond, mask, lohi, shape, noise, branch, fuse, mask_x, record=None, hr=None,
                      return_all_timesteps=False, return_all_outputs=False):
        """-> final tensor (ddpm.py:930-977).  ``record(t, x)`` (optional) sees x_{t-1} after every step.
        ``return_all_timesteps``: torch.stack(imgs, dim=1) with imgs = [x_T, x_{T-1}, ..., x_0] (:946,877-879,963);
        ``return_all_outputs``: (ret, x_start_lst, []) with one x0 per step -- an [out, in] pair for branch steps
        (:869), the fused / single x0 otherwise (:879,922) (:971-972)."""
        o = self.o
        x = noise(shape)
        T = self.T
        if o.start_intermediate and o.use_gt:                # ddpm.py:937-944 (gate: self.start_intermediate, :1099-1102)
            t0 = int(o.use_gt_timestep)
            x = self._c("sqrt_alphas_cumprod", t0) * hr + self._c("sqrt_one_minus_alphas_cumprod", t0) * x   # q_sample, :1148-1154
            T = t0
        imgs, x_start_lst = [x], []
        joint = not branch
        xs = None
        x_branchout = None            # masked branch states kept by the fusion step (ddpm.py:799)
        for t in range(T - 1, -1, -1):
            sigma = (0.5 * self._c("posterior_log_variance_clipped", t)).exp()
            if not joint:
                if xs is None:
                    xs = [x, x]
                if fuse and t <= o.start_timestep:
                    x, x0f, x_branchout = self._fuse_step(xs, cond, mask, t, lohi, mask_x, noise, sigma)
                    self.mask_x_state = False        # ddpm.py:781
                    joint, xs = True, None          # (fusion() only does bookkeeping for this step: branch_cnt == 1, :877)
                    imgs.append(x)
                    x_start_lst.append(x0f)
                    if record:
                        record(t, x)
                    continue
                (_, x0o), (_, x0i) = self.predict_branches(xs[0], xs[1], cond, mask, t, lohi, False, mask_x)
                x0o = x0o.clamp(lohi[0], lohi[1])
                x0i = x0i.clamp(lohi[0], lohi[1])
                z = noise(xs[0].shape) if t > 0 else 0.0
                xs = [self.posterior_mean(x0o, xs[0], t) + sigma * z,
                      self.posterior_mean(x0i, xs[1], t) + sigma * z]
                imgs.append(xs)                      # branching_out(), :865,869
                x_start_lst.append([x0o, x0i])
                if record:
                    record(t, xs)
            else:
                _, x0 = self.predict_single(x, cond, t, lohi, False)
                x0 = x0.clamp(lohi[0], lohi[1])
                z = noise(x.shape) if t > 0 else 0.0
                x = self.posterior_mean(x0, x, t) + sigma * z
                # classifier-gated re-branching (fusion(), ddpm.py:883-916): until a fused prediction is accepted,
                # every joint step is scored; a rejected one (score <= 0, t > 0) is thrown away and replaced by a
                # fresh branch + fusion step at the SAME t from the masked branch states of the fusion time, with
                # mask_x forced on (:906-908).  Noise draws happen in program order (rejected step, then the redo).
                if o.classifier and fuse and o.branch_out and x_branchout is not None:
                    if self.classifier_flag == 0:
                        assert self.classifier is not None, "config['classifier'] needs a callable in .classifier"
                        self.pred_cls = float(self.classifier(x0)[0])
                    if self.pred_cls > 0.0 or t == 0:
                        self.classifier_flag = 1
                    else:
                        mask_x = True                # ddpm.py:908 sets the flag, the redo's fusion step clears it again
                        x, x0, x_branchout = self._fuse_step(x_branchout, cond, mask, t, lohi, mask_x, noise, sigma)
                        self.mask_x_state = False
                imgs.append(x)
                x_start_lst.append(x0)
                if record:
                    record(t, x)
        ret = xs if xs is not None else x
        if return_all_timesteps:
            ret = torch.stack(imgs, dim=1)                   # raises on the [out, in] lists of branch steps, as :963 does
        if (not o.start_intermediate) and o.branch_out:      # ddpm.py:964-970
            ret = torch.stack(ret, dim=0) if isinstance(ret, list) else torch.stack((ret, ret), dim=0)
        if return_all_outputs:
            return ret, x_start_lst, []
        return ret

    # --- K-mask generalisation of the branch -> fusion -> joint loop (NOT in the reference, which has two
    #     branches).  Branch 0 is the OOD-style branch (hard-masked conditioning, mask_x on its prediction), branches
    #     1..K-1 are IND-style (conditioning floored at lo_clip).  Every formula is the reference's with "in" read as
    #     "each further branch": for K = 2 and masks [m, 1 - (m >= 1)] the arithmetic below is operation for operation
    #     that of predict_branches / _fuse_step / p_sample_loop, and tests assert it reproduces the reference goldens
    #     bit for bit.
    def kmask_conditions(self, cond, masks):
        """masks [B,K,H,W] -> (binary masks, [cond_0 .. cond_{K-1}])  (ddpm.py:672-690 per branch)."""
        lo = 0.5 if self.o.data == "mnist" else 0.95
        binary = (masks >= 1.0).float()
        conds = [(cond * binary[:, 0:1]).float()]
        for k in range(1, masks.shape[1]):
            conds.append((cond * torch.clip(binary[:, k:k + 1], lo, 1.0)).float())
        return binary, conds

    def kmask_predict(self, xs, cond, masks, t, lohi, mask_x):
        assert self.o.objective == "pred_x0", "branch mode exists only for pred_x0 (ddpm.py:739-749)"
        binary, conds = self.kmask_conditions(cond, masks)
        tv = self._tvec(xs[0].shape[0], t)
        replaced = mask_x and any(k in self.o.data for k in _REPLACE_OUT) and "mri" not in self.o.data
        outs = [None] + [self.f(xs[k], conds[k], tv) for k in range(1, len(xs))]
        b0 = binary[:, 0:1]
        if replaced:                                             # ddpm.py:704-708
            outs[0] = conds[0].clone()
        else:
            outs[0] = self.f(xs[0], conds[0], tv)
            if mask_x:                                           # ddpm.py:700-703
                outs[0] = outs[0] * b0
                outs[0] = torch.where(b0 == 0.0, torch.tensor(float(lohi[0])), outs[0])
        return binary, [o.clamp(lohi[0], lohi[1]) for o in outs]   # p_mean_variance clamps every branch (:775-776)

    def kmask_fuse(self, xs, x0s, binary, lohi):
        """ddpm.py:784-804 for K branches: x0 = clamp(sum_{k>=1} x0_k m_k + x0_0); x = first non-zero of x_k m_k."""
        acc = x0s[1] * binary[:, 1:2]
        for k in range(2, len(xs)):
            acc = acc + x0s[k] * binary[:, k:k + 1]
        x0 = (acc + x0s[0]).clamp(lohi[0], lohi[1])
        parts = [xs[k] * binary[:, k:k + 1] for k in range(len(xs))]
        x = parts[0]
        for k in range(1, len(xs)):
            x = torch.where(x == 0.0, parts[k], x)
        return x, x0

    def _kmask_fuse_step(self, xs, cond, masks, t, lohi, mask_x, noise, sigma):
        """K-branch evaluation + fusion of one step -> (x_{t-1}, x0, [x_k m_k]): _fuse_step for K branches."""
        binary, x0s = self.kmask_predict(xs, cond, masks, t, lohi, mask_x)
        x, x0 = self.kmask_fuse(xs, x0s, binary, lohi)
        kept = [xs[k] * binary[:, k:k + 1] for k in range(len(xs))]
        z = noise(x.shape) if t > 0 else 0.0
        return self.posterior_mean(x0, x, t) + sigma * z, x0, kept

    def p_sample_loop_kmask(self, cond, masks, lohi, shape, noise, fuse, mask_x, record=None, return_all_outputs=False):
        """K branches with a shared draw per step (ddpm.py:852-858), fused at t <= start_timestep when ``fuse``, joint
        single-branch steps afterwards; without ``fuse`` returns the K branch states stacked [K,B,C,H,W].
        ``return_all_outputs``: (ret, x_start_lst, []) as ddpm.py:971-972 -- a K-list of x0 per branch step (:869), the
        fused / single x0 otherwise.  With o.classifier the joint steps run under the gate of fusion() (:883-916): a
        rejected step is replaced by a K-branch evaluation + fusion at the same t from the masked branch states the
        fusion step kept, with mask_x forced on -- the two-branch rule with "each further branch" for "in"."""
        o, K = self.o, masks.shape[1]
        x = noise(shape)
        xs = [x] * K
        joint = False
        x_start_lst, kept = [], None
        for t in range(self.T - 1, -1, -1):
            sigma = (0.5 * self._c("posterior_log_variance_clipped", t)).exp()
            if not joint:
                if fuse and t <= o.start_timestep:
                    x, x0, kept = self._kmask_fuse_step(xs, cond, masks, t, lohi, mask_x, noise, sigma)
                    self.mask_x_state = False
                    joint, xs = True, None
                    x_start_lst.append(x0)
                    if record:
                        record(t, x)
                    continue
                binary, x0s = self.kmask_predict(xs, cond, masks, t, lohi, mask_x)
                z = noise(xs[0].shape) if t > 0 else 0.0
                xs = [self.posterior_mean(x0s[k], xs[k], t) + sigma * z for k in range(K)]
                x_start_lst.append(list(x0s))
                if record:
                    record(t, xs)
            else:
                _, x0 = self.predict_single(x, cond, t, lohi, False)
                x0 = x0.clamp(lohi[0], lohi[1])
                z = noise(x.shape) if t > 0 else 0.0
                x = self.posterior_mean(x0, x, t) + sigma * z
                if o.classifier and fuse and kept is not None:
                    if self.classifier_flag == 0:
                        assert self.classifier is not None, "config['classifier'] needs a callable in .classifier"
                        self.pred_cls = float(self.classifier(x0)[0])
                    if self.pred_cls > 0.0 or t == 0:
                        self.classifier_flag = 1
                    else:
                        mask_x = True
                        x, x0, kept = self._kmask_fuse_step(kept, cond, masks, t, lohi, mask_x, noise, sigma)
                        self.mask_x_state = False
                x_start_lst.append(x0)
                if record:
                    record(t, x)
        ret = x if joint else torch.stack(xs, dim=0)
        if return_all_outputs:
            return ret, x_start_lst, []
        return ret

    def ddim_sample_kmask(self, cond, masks, lohi, shape, noise, fuse, mask_x):
        """ddim_sample (ddpm.py:980-1075) for K branches.  Per pair every branch is evaluated with clip_x_start
        (:1005), eps re-derived from the clamped x0 (:745-746), one shared draw (:1020).  Fusion at t <= times[-s-2]
        (:1022-1041 read per branch): x0 = clamp(x0_0 where it is non-zero, else the x0 of the IND branch that owns
        the pixel -- the first k >= 1 with m_k >= 1, the last branch if none does); eps = first non-zero of
        eps_k m_k (the last one if all are zero).  For K = 2 and m_1 = 1 - (m_0 >= 1) both rules are the reference's
        where(x0_out == 0, x0_in, x0_out) and where(eps_out m == 0, eps_in (1 - m), eps_out m).  Never fused: the K
        branch states come back as a list."""
        o, K = self.o, masks.shape[1]
        times = list(reversed(torch.linspace(-1, self.T - 1, steps=self.S + 1).int().tolist()))
        pairs = list(zip(times[:-1], times[1:]))
        t_fuse = times[-o.start_timestep - 2]
        eta, abar = o.ddim_sampling_eta, self.buf["alphas_cumprod"]
        x = noise(shape)
        xs, joint = [x] * K, False
        for t, t_next in pairs:
            if not joint:
                binary, x0s = self.kmask_predict(xs, cond, masks, t, lohi, mask_x)      # clamped per branch
                eps = [self.eps_from_x0(xs[k], t, x0s[k]) for k in range(K)]
                if t_next < 0:
                    xs = x0s
                    continue
                a, an = abar[t], abar[t_next]
                sigma = eta * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
                c = (1 - an - sigma ** 2).sqrt()
                z = noise(xs[0].shape)
                if fuse and t <= t_fuse:
                    alt = x0s[K - 1]
                    for k in range(K - 2, 0, -1):
                        alt = torch.where(binary[:, k:k + 1] >= 1.0, x0s[k], alt)
                    x0 = torch.where(x0s[0] == 0.0, alt, x0s[0]).clamp(lohi[0], lohi[1])
                    e = eps[0] * binary[:, 0:1]
                    for k in range(1, K):
                        e = torch.where(e == 0.0, eps[k] * binary[:, k:k + 1], e)
                    x = x0 * an.sqrt() + c * e + sigma * z
                    self.mask_x_state = False
                    joint, xs = True, None
                else:
                    xs = [x0s[k] * an.sqrt() + c * eps[k] + sigma * z for k in range(K)]
            else:
                e, x0 = self.predict_single(x, cond, t, lohi, True)
                if o.branch_out:
                    x0 = x0.clamp(lohi[0], lohi[1])
                if t_next < 0:
                    x = x0
                    continue
                a, an = abar[t], abar[t_next]
                sigma = eta * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
                c = (1 - an - sigma ** 2).sqrt()
                x = x0 * an.sqrt() + c * e + sigma * noise(x.shape)
        return x if joint else list(xs)

    # --- DDIM (ddpm.py:980-1075) ---
    def ddim_sample(self, cond, mask, lohi, shape, noise, branch, fuse, mask_x, return_all_timesteps=False):
        o = self.o
        times = torch.linspace(-1, self.T - 1, steps=self.S + 1)
        times = list(reversed(times.int().tolist()))
        pairs = list(zip(times[:-1], times[1:]))
        t_fuse = times[-o.start_timestep - 2]
        eta = o.ddim_sampling_eta
        abar = self.buf["alphas_cumprod"]
        x = noise(shape)
        imgs = [x, x] if branch else [x]                      # ddpm.py:990-993
        joint = not branch
        xs = None
        for t, t_next in pairs:
            if not joint:
                if xs is None:
                    xs = [x, x]
                (eo, x0o), (ei, x0i) = self.predict_branches(xs[0], xs[1], cond, mask, t, lohi, True, mask_x)
                if t_next < 0:
                    xs = [x0o, x0i]
                    imgs.append(xs)
                    continue
                a, an = abar[t], abar[t_next]
                sigma = eta * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
                c = (1 - an - sigma ** 2).sqrt()
                z = noise(xs[0].shape)
                if fuse and t <= t_fuse:
                    x0 = torch.where(x0o == 0.0, x0i, x0o).clamp(lohi[0], lohi[1])
                    m = (mask >= 1.0).float()
                    po, pi = eo * m, ei * (1.0 - m)
                    assert bool((po == 0).any()) and bool((pi == 0).any()), "x_out and x_in should be masked"
                    eps = torch.where(po == 0.0, pi, po)
                    x = x0 * an.sqrt() + c * eps + sigma * z
                    self.mask_x_state = False        # ddpm.py:1024
                    joint, xs = True, None
                    imgs.append(x)
                else:
                    xs = [x0o * an.sqrt() + c * eo + sigma * z, x0i * an.sqrt() + c * ei + sigma * z]
                    imgs.append(xs)
            else:
                eps, x0 = self.predict_single(x, cond, t, lohi, True)
                if o.branch_out:
                    x0 = x0.clamp(lohi[0], lohi[1])
                if t_next < 0:
                    x = x0
                    imgs.append(x)
                    continue
                a, an = abar[t], abar[t_next]
                sigma = eta * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
                c = (1 - an - sigma ** 2).sqrt()
                z = noise(x.shape)
                x = x0 * an.sqrt() + c * eps + sigma * z
                imgs.append(x)
        if return_all_timesteps:
            return torch.stack(imgs, dim=1)                   # ddpm.py:1072 (raises on per-branch lists, as there)
        return xs if xs is not None else x

    # --- dispatcher (ddpm.py:1078-1125) ---
    def sample(self, cond, mask, lohi, batch_size, noise, gt=None, return_all_timesteps=False, return_all_outputs=False):
        """One call of the reference's sample().  branch_out / start_intermediate are restored from the constructor
        copies at every call (:1093-1097); config['mask_x'] is NOT -- it carries over (``mask_x_state``)."""
        o = self.o
        branch, fuse = o.branch_out, o.start_intermediate
        if o.ood_AD or o.ood_confidence:               # :1106-1108
            self.mask_x_state = True
        mask_x = self.mask_x_state
        if branch and mask is not None:
            u = torch.unique(mask)
            if len(u) == 1 and float(u[0]) == 1.0:     # all-ones mask: plain reverse process (:1110-1117)
                branch, fuse, mask_x = False, False, False
                self.mask_x_state = False              # :1114, never restored
        shape = (batch_size, self.channels, self.image_size, self.image_size)
        if self.S < self.T:
            return self.ddim_sample(cond, mask, lohi, shape, noise, branch, fuse, mask_x,
                                    return_all_timesteps=return_all_timesteps)
        return self.p_sample_loop(cond, mask, lohi, shape, noise, branch, fuse, mask_x, hr=gt,
                                  return_all_timesteps=return_all_timesteps, return_all_outputs=return_all_outputs)


def make_model_fn(sd, cfg):
    """Bind oracle/unet_ref.unet_forward to a weight dict."""
    def f(x, cond, t):
        with torch.no_grad():
            return unet_ref.unet_forward(sd, cfg, x, cond, t)
    return f
