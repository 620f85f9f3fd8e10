"""TEST INFRASTRUCTURE: a stand-in for storygen_amd.engine.UNetEngine + the three loop-level ops the sampler calls, in plain torch
on the CPU.  The arithmetic is arbitrary but deterministic and sensitive to everything the SCHEDULE can get wrong: which context
set a main pass reads, which parameter row, which reference timestep / noise a sample got, where a harvested feature landed, and
whether the K / V^T buffers belong to the context set they are used with.  It lets the sampler's schedules (step by step, look-ahead,
groups of G steps) be compared with each other — and bench.py's launcher path be exercised — without a GPU.  It says nothing about
the kernels."""
import torch

KEYS = ("lo", "hi")
C = 4


class StubEngine:
    def __init__(self, arch, state_dict, device, batch, height, width, n_ref=0, seq_len=77, weights=None, ctx_rows=None,
                 attn3_groups=None, fp8_attention=False, ctx_short=0, cfg_shared_head=False, **_):
        f32 = torch.float32
        self.B, self.R, self.hw = batch, n_ref, 2
        self.x_in = torch.zeros(batch, 4, height, width, dtype=f32)
        self.t_in = torch.zeros(batch, dtype=f32)
        self.text_in = torch.zeros(batch, seq_len, 8, dtype=torch.float16)
        self.eps_out = torch.zeros(batch, 4, height, width, dtype=f32)
        self.ctx_rows = batch if ctx_rows is None else ctx_rows
        self.ctx_short = int(ctx_short)
        self.ctx_slots = self.ctx_short + (self.ctx_rows - self.ctx_short) * n_ref
        self.attn3_groups = [(0, batch, 0)] if attn3_groups is None else list(attn3_groups)
        self.ctx, self.kv_ext = {}, None
        if n_ref:
            for k in KEYS:
                self.ctx[k] = (torch.zeros(self.ctx_slots * self.hw, C) if self.ctx_short
                               else torch.zeros(self.ctx_rows, n_ref * self.hw, C))
        self.calls = []

    def cache_text_kv(self):
        pass

    def build_time_table(self, timesteps):
        return False

    def _feature(self, key, u):
        base = 1.0 if key == "lo" else -2.0
        v = base * self.x_in[u].mean() + 1e-3 * self.t_in[u] + self.text_in[u].float().mean()
        return v + torch.arange(self.hw * C, dtype=torch.float32).view(self.hw, C) * 1e-2

    def forward(self, harvest=None, harvest_only=False, consume=False, text_cache=False, side=None, **_):
        hw = self.hw
        if harvest is not None:
            plans = list(harvest) if isinstance(harvest, (list, tuple)) else [harvest]
            self.calls.append(("ref", self.B))
            for key in KEYS:
                for plan in plans:
                    c2d = plan.ctx[key].view(-1, C)
                    if plan.identity:
                        assert c2d.shape[0] == self.B * hw
                        for u in range(self.B):
                            c2d[u * hw:(u + 1) * hw] = self._feature(key, u)
                    else:
                        R = plan.slots_per_row or (plan.ctx[key].shape[1] // hw)
                        for src, step, row, slot, cnt in plan.ops:
                            for j in range(cnt):
                                f = plan.flat_slot(row, slot + j, R)
                                c2d[f * hw:(f + 1) * hw] = self._feature(key, plan.src_offset + src + j * step)
                    if plan.kv is not None:
                        ki, vti = plan.kv[key]
                        ki.copy_(2.0 * c2d)
                        vti.copy_((3.0 * c2d).t())
            return None
        self.calls.append(("main", self.B))
        eps = 0.1 * self.x_in + 1e-4 * self.t_in.view(-1, 1, 1, 1) + self.text_in.float().mean(dim=(1, 2)).view(-1, 1, 1, 1)
        if consume:
            ns, rows, R = self.ctx_short, self.ctx_rows, self.R
            for key in KEYS:
                c2d = self.ctx[key].view(-1, C)
                ki, vti = self.kv_ext[key]
                assert torch.equal(ki, 2.0 * c2d) and torch.equal(vti, (3.0 * c2d).t()), "K / V^T do not belong to this context set"
                for q0, n, c0 in self.attn3_groups:
                    for i in range(n):
                        row = c0 + i
                        lo = row * hw if row < ns else (ns + (row - ns) * R) * hw
                        cnt = hw if row < ns else R * hw
                        w = torch.arange(1, cnt + 1, dtype=torch.float32).view(-1, 1) / cnt       # slot order matters
                        eps[q0 + i] += (ki[lo:lo + cnt] * w).mean() * (1.0 if key == "lo" else 0.5)
        self.eps_out.copy_(eps)
        return self.eps_out


def add_noise(src, noise, coef, out):
    out.copy_(coef[:, 0].view(-1, 1, 1, 1) * src + coef[:, 1].view(-1, 1, 1, 1) * noise)
    return out


def cfg_ddim_step(eps3, latents, latents3, coef):
    n = latents.shape[0]
    e0, e1, e2 = eps3[:n], eps3[n:2 * n], eps3[2 * n:]
    eps = e0 + coef[0] * (e1 - e0) + coef[1] * (e2 - e1)
    latents.copy_(coef[2] * latents + coef[3] * eps + coef[4] * 1e-3 + coef[5] * 1e-3)
    if latents3 is not None:
        latents3.copy_(torch.cat([latents] * 3))
    return latents


def install(monkeypatch):
    """Point storygen_amd.sampler at the stand-ins."""
    import types

    import storygen_amd.sampler as S
    monkeypatch.setattr(S, "UNetEngine", StubEngine)
    monkeypatch.setattr(S, "ops", types.SimpleNamespace(add_noise=add_noise, cfg_ddim_step=cfg_ddim_step))
    return S
