"""Pins oracle/s3gen_oracle.py op-for-op against CPU libtorch (python `torch`), the way tests/test_oracle_vs_torch.py pins the
T3 oracle: the same sequence as R/s3gen.R:64-528 written with the ATen entry points R torch dispatches to (conv1d on a
left-padded input, layer_norm, softplus + tanh, gelu(approximate = "tanh"), softmax, matmul), on a synthetic checkpoint in
the key layout of load_cfm_estimator_weights (R/s3gen.R:1070-1178).  The reference holds no S3Gen fixture and its weights
are not in the container: parity against the reference itself is UNPINNED (stated in the oracle's header and DESIGN.md)."""
import numpy as np
import pytest

from chatterbox_amd import synth
from oracle import s3gen_oracle as S

torch = pytest.importorskip("torch")
F = torch.nn.functional


class TorchEstimator:
    def __init__(self, w, dims, prefix=synth.CFM_PREFIX):
        self.w = {k[len(prefix):]: torch.from_numpy(v) for k, v in w.items()}
        self.d = dims

    def temb(self, t):
        half = self.d.time_in // 2
        emb = torch.exp(torch.arange(half, dtype=torch.float32) * (-np.log(10000.0) / (half - 1)))
        e = 1000 * t.unsqueeze(1) * emb.unsqueeze(0)
        e = torch.cat([e.sin(), e.cos()], dim=-1)
        w = self.w
        e = F.linear(e, w["time_mlp.linear_1.weight"], w["time_mlp.linear_1.bias"])
        return F.linear(F.silu(e), w["time_mlp.linear_2.weight"], w["time_mlp.linear_2.bias"])

    def cconv(self, x, name):
        k = self.w[name + ".weight"].shape[2]
        return F.conv1d(F.pad(x, (k - 1, 0)), self.w[name + ".weight"], self.w[name + ".bias"])

    def block(self, x, mask, p):
        h = self.cconv(x * mask, p + "block.0").transpose(1, 2)
        h = F.layer_norm(h, (h.shape[-1],), self.w[p + "block.2.weight"], self.w[p + "block.2.bias"]).transpose(1, 2)
        return h * torch.tanh(F.softplus(h)) * mask

    def resnet(self, x, mask, te, p):
        h = self.block(x, mask, p + "block1.")
        m = te * torch.tanh(F.softplus(te))
        h = h + F.linear(m, self.w[p + "mlp.1.weight"], self.w[p + "mlp.1.bias"]).unsqueeze(-1)
        h = self.block(h, mask, p + "block2.")
        return h + F.conv1d(x * mask, self.w[p + "res_conv.weight"], self.w[p + "res_conv.bias"])

    def tfm(self, x, p):
        w, H, Dh = self.w, self.d.heads, self.d.head_dim
        B, T, C = x.shape
        n = F.layer_norm(x, (C,), w[p + "norm1.weight"], w[p + "norm1.bias"])
        q, k, v = (F.linear(n, w[p + f"attn1.to_{c}.weight"]).view(B, T, H, Dh).transpose(1, 2) for c in "qkv")
        a = torch.softmax(torch.matmul(q, k.transpose(-2, -1)) * Dh ** -0.5, dim=-1)
        o = torch.matmul(a, v).transpose(1, 2).contiguous().view(B, T, -1)
        x = x + F.linear(o, w[p + "attn1.to_out.0.weight"], w[p + "attn1.to_out.0.bias"])
        n = F.layer_norm(x, (C,), w[p + "norm3.weight"], w[p + "norm3.bias"])
        f = F.gelu(F.linear(n, w[p + "ff.net.0.proj.weight"], w[p + "ff.net.0.proj.bias"]), approximate="tanh")
        return x + F.linear(f, w[p + "ff.net.2.weight"], w[p + "ff.net.2.bias"])

    def forward(self, x, mask, mu, t, spks, cond, r=None):
        te = self.temb(t)
        if r is not None:
            te = F.linear(torch.cat([te, self.temb(r)], dim=1), self.w["time_embed_mixer.weight"])
        T = x.shape[2]
        h = torch.cat([x, mu, spks.unsqueeze(2).expand(-1, -1, T), cond], dim=1)
        h = self.resnet(h, mask, te, "down_blocks.0.0.").transpose(1, 2)
        for i in range(self.d.n_tfm):
            h = self.tfm(h, f"down_blocks.0.1.{i}.")
        h = h.transpose(1, 2)
        skip = h
        h = self.cconv(h * mask, "down_blocks.0.2")
        for m in range(self.d.n_mid):
            h = self.resnet(h, mask, te, f"mid_blocks.{m}.0.").transpose(1, 2)
            for j in range(self.d.n_tfm):
                h = self.tfm(h, f"mid_blocks.{m}.1.{j}.")
            h = h.transpose(1, 2)
        h = self.resnet(torch.cat([h, skip], dim=1), mask, te, "up_blocks.0.0.").transpose(1, 2)
        for i in range(self.d.n_tfm):
            h = self.tfm(h, f"up_blocks.0.1.{i}.")
        h = self.cconv(h.transpose(1, 2) * mask, "up_blocks.0.2")
        h = self.block(h, mask, "final_block.")
        return F.conv1d(h * mask, self.w["final_proj.weight"], self.w["final_proj.bias"])


def _case(dims, T, seed, B=2):
    w = synth.synth_cfm_weights(dims, 777)
    mu, spks, cond, z = synth.synth_cfm_inputs(dims, T, seed, B)
    mask = np.ones((B, 1, T), np.float32)
    return w, mu, spks, cond, z, mask


@pytest.mark.parametrize("dims,T", [(synth.CFM_TINY, 37), (synth.CfmDims(n_mid=1, n_tfm=1), 50)], ids=["tiny", "full-width"])
def test_estimator_matches_libtorch(dims, T):
    w, mu, spks, cond, z, mask = _case(dims, T, 3)
    mask[1, 0, T - 5:] = 0.0                     # a padded tail in row 2: the mask multiplies, attention stays unmasked
    t = np.array([0.3, 0.85], np.float32)
    got = S.CfmEstimatorOracle(w, heads=dims.heads, head_dim=dims.head_dim, time_in=dims.time_in).forward(z, mask, mu, t, spks, cond)
    te = TorchEstimator(w, dims)
    with torch.no_grad():
        want = te.forward(*(torch.from_numpy(a) for a in (z, mask, mu, t, spks, cond))).numpy()
    assert got.shape == (2, dims.mel, T)
    assert np.abs(got - want).max() < 2e-5 * max(1.0, np.abs(want).max())


def test_meanflow_mixer_matches_libtorch():
    dims = synth.CfmDims(mel=16, hidden=64, n_mid=1, n_tfm=1, heads=2, time_in=64, time_dim=128, meanflow=True)
    w, mu, spks, cond, z, mask = _case(dims, 21, 4, B=1)
    t, r = np.array([0.0], np.float32), np.array([0.5], np.float32)
    est = S.CfmEstimatorOracle(w, heads=2, head_dim=64, time_in=64)
    assert est.meanflow
    got = est.forward(z, mask, mu, t, spks, cond, r=r)
    with torch.no_grad():
        want = TorchEstimator(w, dims).forward(*(torch.from_numpy(a) for a in (z, mask, mu, t, spks, cond)), r=torch.from_numpy(r)).numpy()
    assert np.abs(got - want).max() < 2e-5 * max(1.0, np.abs(want).max())


def test_euler_solver_matches_libtorch_loop():
    """solve_euler (R/s3gen.R:651-751) restated in torch beside the oracle's: cosine schedule, CFG batch of two with
    the second row's mu / spks / cond zeroed, (1 + 0.7) cond - 0.7 uncond, t accumulated step by step."""
    dims = synth.CFM_TINY
    w, mu, spks, cond, z, mask = _case(dims, 29, 9, B=1)
    est = S.CfmEstimatorOracle(w, heads=dims.heads, head_dim=dims.head_dim, time_in=dims.time_in)
    got = S.solve_euler(est, z, mu, mask, spks, cond, n_timesteps=10)
    te = TorchEstimator(w, dims)
    with torch.no_grad():
        tz, tmu, tmask, tspk, tcond = (torch.from_numpy(a) for a in (z, mu, mask, spks, cond))
        ts = torch.linspace(0, 1, 11, dtype=torch.float32)
        ts = 1 - torch.cos(ts * 0.5 * np.pi)
        x, t, dt = tz, ts[0].unsqueeze(0), ts[1] - ts[0]
        for step in range(1, 11):
            x_in = torch.cat([x, x]); m_in = torch.cat([tmask, tmask])
            mu_in = torch.cat([tmu, torch.zeros_like(tmu)]); s_in = torch.cat([tspk, torch.zeros_like(tspk)])
            c_in = torch.cat([tcond, torch.zeros_like(tcond)]); t_in = torch.cat([t, t])
            d = te.forward(x_in, m_in, mu_in, t_in, s_in, c_in)
            x = x + dt * ((1.0 + 0.7) * d[0:1] - 0.7 * d[1:2])
            t = t + dt
            if step < 10:
                dt = ts[step + 1] - ts[step]
        want = x.numpy()
    assert np.isfinite(got).all() and np.abs(got - want).max() < 5e-5 * max(1.0, np.abs(want).max())
    # the schedule itself: 0 at the start, 1 at the end, cosine spacing (finer steps first)
    for n in (1, 2, 3, 7, 10, 32):     # the float32 linspace itself, bit for bit
        assert np.array_equal(S.linspace_f32(n), torch.linspace(0, 1, n + 1, dtype=torch.float32).numpy()), n
    sp = S.t_span_cosine(10)
    assert sp[0] == 0 and abs(sp[-1] - 1) < 1e-6 and np.all(np.diff(sp) > 0) and sp[1] - sp[0] < sp[-1] - sp[-2]


def test_synthetic_checkpoint_has_the_loader_layout():
    """Every key load_cfm_estimator_weights copies (R/s3gen.R:1092-1175) exists with the module's shape."""
    d = synth.CFM_FULL
    w = synth.synth_cfm_weights(d, 777)
    p = synth.CFM_PREFIX
    assert w[p + "time_mlp.linear_1.weight"].shape == (1024, 320)
    assert w[p + "down_blocks.0.0.block1.block.0.weight"].shape == (256, 320, 3)
    assert w[p + "down_blocks.0.0.res_conv.weight"].shape == (256, 320, 1)
    assert w[p + "up_blocks.0.0.block1.block.0.weight"].shape == (256, 512, 3)
    assert w[p + "mid_blocks.11.1.3.attn1.to_q.weight"].shape == (512, 256)
    assert w[p + "mid_blocks.11.1.3.attn1.to_out.0.bias"].shape == (256,)
    assert w[p + "mid_blocks.0.1.0.ff.net.0.proj.weight"].shape == (1024, 256)
    assert w[p + "final_proj.weight"].shape == (80, 256, 1)
    n_tfm = sum(1 for k in w if k.endswith("norm1.weight"))
    assert n_tfm == 56 and sum(v.size for v in w.values()) > 60e6


# ---- conformer encoder + flow module (R/conformer.R, R/s3gen.R:805-890) ------------------------------------------------

class TorchConformer:
    """upsample_conformer_encoder_full$forward written with the ATen calls of R/conformer.R, B = 1, no padding."""

    def __init__(self, w, dims, prefix="flow.encoder."):
        self.w = {k[len(prefix):]: torch.from_numpy(v) for k, v in w.items() if k.startswith(prefix)}
        self.d, self.H = dims.d, dims.heads

    def pe(self, T, max_len=700):
        d = self.d
        position = torch.arange(0, max_len, dtype=torch.float32).unsqueeze(1)
        div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * (-(np.log(10000.0) / d)))
        pos = torch.zeros(max_len, d); neg = torch.zeros(max_len, d)
        pos[:, 0::2] = torch.sin(position * div); pos[:, 1::2] = torch.cos(position * div)
        neg[:, 0::2] = torch.sin(-1 * position * div); neg[:, 1::2] = torch.cos(-1 * position * div)
        pe = torch.cat([torch.flip(pos, [0]), neg[1:]], dim=0).unsqueeze(0)       # R/conformer.R:58-69
        center = max_len                                                           # 1-indexed centre (:83)
        return pe[:, center - T:center + T - 1]                                    # rows start_idx..end_idx, 1-indexed inclusive

    def embed(self, x, p):
        w = self.w
        h = F.layer_norm(F.linear(x, w[p + "out.0.weight"], w[p + "out.0.bias"]), (self.d,), w[p + "out.1.weight"], w[p + "out.1.bias"])
        return h * np.sqrt(self.d), self.pe(x.shape[1])

    def rel_shift(self, x):
        b, h, t, p = x.shape
        x = F.pad(x, (1, 0)).view(b, h, p + 1, t)[:, :, 1:, :]
        return x.reshape(b, h, t, p)[:, :, :, :t].contiguous()

    def attn(self, x, pos, p):
        w, H = self.w, self.H
        B, T, d = x.shape
        dk = d // H
        q, k, v = (F.linear(x, w[p + f"linear_{c}.weight"], w[p + f"linear_{c}.bias"]).view(B, -1, H, dk).transpose(1, 2) for c in "qkv")
        pp = F.linear(pos, w[p + "linear_pos.weight"]).view(1, -1, H, dk).transpose(1, 2)
        qu = q + w[p + "pos_bias_u"].unsqueeze(0).unsqueeze(2)
        qv = q + w[p + "pos_bias_v"].unsqueeze(0).unsqueeze(2)
        s = (torch.matmul(qu, k.transpose(-2, -1)) + self.rel_shift(torch.matmul(qv, pp.transpose(-2, -1)))) / np.sqrt(dk)
        o = torch.matmul(torch.softmax(s, dim=-1), v).transpose(1, 2).contiguous().view(B, -1, d)
        return F.linear(o, w[p + "linear_out.weight"], w[p + "linear_out.bias"])

    def layer(self, x, pos, p):
        w = self.w
        n = F.layer_norm(x, (self.d,), w[p + "norm_mha.weight"], w[p + "norm_mha.bias"], eps=1e-12)
        x = x + self.attn(n, pos, p + "self_attn.")
        n = F.layer_norm(x, (self.d,), w[p + "norm_ff.weight"], w[p + "norm_ff.bias"], eps=1e-12)
        return x + F.linear(F.silu(F.linear(n, w[p + "feed_forward.w_1.weight"], w[p + "feed_forward.w_1.bias"])),
                            w[p + "feed_forward.w_2.weight"], w[p + "feed_forward.w_2.bias"])

    def forward(self, x, n_blocks, n_up):
        w = self.w
        xs, pos = self.embed(x, "embed.")
        h = F.pad(xs.transpose(1, 2), (0, 3))
        h = F.leaky_relu(F.conv1d(h, w["pre_lookahead_layer.conv1.weight"], w["pre_lookahead_layer.conv1.bias"]))
        h = F.conv1d(F.pad(h, (2, 0)), w["pre_lookahead_layer.conv2.weight"], w["pre_lookahead_layer.conv2.bias"])
        xs = h.transpose(1, 2) + xs
        for i in range(n_blocks):
            xs = self.layer(xs, pos, f"encoders.{i}.")
        h = F.interpolate(xs.transpose(1, 2), scale_factor=2.0, mode="nearest")
        h = F.conv1d(F.pad(h, (4, 0)), w["up_layer.conv.weight"], w["up_layer.conv.bias"])
        xs, pos = self.embed(h.transpose(1, 2), "up_embed.")
        for i in range(n_up):
            xs = self.layer(xs, pos, f"up_encoders.{i}.")
        return F.layer_norm(xs, (self.d,), w["after_norm.weight"], w["after_norm.bias"])


@pytest.mark.parametrize("dims,T", [(synth.FLOW_TINY, 23), (synth.FlowDims(n_blocks=1, n_up=1, cfm=synth.CFM_TINY), 31)], ids=["tiny", "full-width"])
def test_conformer_encoder_matches_libtorch(dims, T):
    w = synth.synth_flow_weights(dims, 555)
    x = np.random.default_rng(2).standard_normal((T, dims.d)).astype(np.float32)
    got = S.ConformerOracle(w, heads=dims.heads).forward(x)
    with torch.no_grad():
        want = TorchConformer(w, dims).forward(torch.from_numpy(x)[None], dims.n_blocks, dims.n_up)[0].numpy()
    assert got.shape == (2 * T, dims.d)
    assert np.abs(got - want).max() < 3e-5 * max(1.0, np.abs(want).max())


def test_rel_shift_picks_the_relative_position():
    """After rel_shift entry (i, j) is column T - 1 - i + j of the raw product: the row of relative position i - j."""
    T = 7
    bd = np.arange(2 * T * (2 * T - 1), dtype=np.float32).reshape(2, T, 2 * T - 1)
    sh = S.rel_shift(bd)
    for i in range(T):
        for j in range(T):
            assert sh[1, i, j] == bd[1, i, T - 1 - i + j]


def test_flow_inputs_and_inference_shapes():
    """causal_masked_diff_xvec$forward (R/s3gen.R:805-890) on the tiny model: mel_len1 = 2 x prompt tokens, the conditioning
    holds the prompt mel over those frames and zeros after, the result is the generated part only."""
    dims = synth.FLOW_TINY
    w = synth.synth_flow_weights(dims, 555)
    tok, ptok, pfeat, xvec, z = synth.synth_flow_inputs(dims, 9, 6, 4)
    mu, spks, cond, l1 = S.flow_inputs(w, tok, ptok, pfeat, xvec, heads=dims.heads)
    assert l1 == 12 and mu.shape == (1, dims.cfm.mel, 30) and spks.shape == (1, dims.cfm.mel)
    assert np.array_equal(cond[0, :, :12], pfeat.T) and not cond[0, :, 12:].any()
    with torch.no_grad():
        e = torch.from_numpy(xvec)[None]
        want = F.linear(F.normalize(e, dim=1), torch.from_numpy(w["flow.spk_embed_affine_layer.weight"]),
                        torch.from_numpy(w["flow.spk_embed_affine_layer.bias"])).numpy()
    assert np.abs(spks - want).max() < 1e-6
    out = S.flow_inference(w, tok, ptok, pfeat, xvec, z, n_timesteps=3, heads=dims.heads, cfm_heads=dims.cfm.heads)
    assert out.shape == (1, dims.cfm.mel, 18) and np.isfinite(out).all()


# ---- HiFT vocoder (R/hifigan.R) ---------------------------------------------------------------------------------------

class TorchHift:
    """hift_generator$inference with the ATen calls of R/hifigan.R, weights through load_hifigan_weights' weight-norm fold."""

    def __init__(self, w, prefix="mel2wav."):
        raw = {k[len(prefix):]: torch.from_numpy(v) for k, v in w.items() if k.startswith(prefix)}
        self.w = {}
        for k, v in raw.items():
            if k.endswith(".parametrizations.weight.original0"):
                base = k[: -len(".parametrizations.weight.original0")]
                vv = raw[base + ".parametrizations.weight.original1"]
                nrm = vv.reshape(vv.shape[0], -1).norm(dim=1, keepdim=True).unsqueeze(2)
                self.w[base + ".weight"] = v * vv / nrm
            elif not k.endswith(".parametrizations.weight.original1"):
                self.w[k] = v
        self.window = torch.tensor(0.5 * (1 - np.cos(2 * np.pi * np.arange(16) / 16)), dtype=torch.float32)

    def snake(self, x, a):
        a = a.unsqueeze(0).unsqueeze(2)
        return x + (1.0 / (a + 1e-9)) * torch.pow(torch.sin(x * a), 2)

    def resblock(self, x, p, k):
        for j, d in enumerate((1, 3, 5)):
            xt = self.snake(x, self.w[f"{p}.activations1.{j}.alpha"])
            xt = F.conv1d(xt, self.w[f"{p}.convs1.{j}.weight"], self.w[f"{p}.convs1.{j}.bias"], dilation=d, padding=(k * d - d) // 2)
            xt = self.snake(xt, self.w[f"{p}.activations2.{j}.alpha"])
            xt = F.conv1d(xt, self.w[f"{p}.convs2.{j}.weight"], self.w[f"{p}.convs2.{j}.bias"], padding=(k - 1) // 2)
            x = xt + x
        return x

    def f0(self, mel):
        x = mel
        for i in range(5):
            x = F.elu(F.conv1d(x, self.w[f"f0_predictor.condnet.{2 * i}.weight"], self.w[f"f0_predictor.condnet.{2 * i}.bias"], padding=1))
        return torch.abs(F.linear(x.transpose(1, 2), self.w["f0_predictor.classifier.weight"], self.w["f0_predictor.classifier.bias"]).squeeze(2))

    def source(self, f0, phase, noise):
        s = F.interpolate(f0.unsqueeze(1), scale_factor=480.0, mode="nearest")            # (1, 1, T_wav)
        nh = self.w["m_source.l_linear.weight"].shape[1]
        fm = torch.zeros(1, nh, s.shape[2])
        for i in range(nh):
            fm[:, i, :] = s.squeeze(1) * (i + 1) / 24000
        theta = 2 * np.pi * (torch.cumsum(fm, dim=2) % 1)
        ph = phase.clone().view(1, nh, 1)
        ph[:, 0, :] = 0
        sine = 0.1 * torch.sin(theta + ph)
        uv = (s > 10).to(torch.float32)
        namp = uv * 0.003 + (1 - uv) * 0.1 / 3
        sine = sine * uv + namp * noise.unsqueeze(0)
        return torch.tanh(F.linear(sine.transpose(1, 2), self.w["m_source.l_linear.weight"], self.w["m_source.l_linear.bias"])).transpose(1, 2)

    def decode(self, mel, s):
        w = self.w
        spec = torch.stft(s.squeeze(1), n_fft=16, hop_length=4, win_length=16, window=self.window, return_complex=True)
        s_stft = torch.cat([spec.real, spec.imag], dim=1)
        x = F.conv1d(mel, w["conv_pre.weight"], w["conv_pre.bias"], padding=3)
        ups, total, cum = ((8, 16), (5, 11), (3, 7)), 120, 1
        for i, (u, k) in enumerate(ups):
            x = F.conv_transpose1d(F.leaky_relu(x, 0.1), w[f"ups.{i}.weight"], w[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
            if i == 2:
                x = F.pad(x, (1, 0), mode="reflect")
            cum *= u
            du = total // cum
            si = F.conv1d(s_stft, w[f"source_downs.{i}.weight"], w[f"source_downs.{i}.bias"], stride=du, padding=0 if du == 1 else du // 2)
            si = self.resblock(si, f"source_resblocks.{i}", (7, 7, 11)[i])
            L = min(x.shape[2], si.shape[2])
            x = x[:, :, :L] + si[:, :, :L]
            xs = sum(self.resblock(x, f"resblocks.{i * 3 + j}", rk) for j, rk in enumerate((3, 7, 11)))
            x = xs / 3
        x = F.conv1d(F.leaky_relu(x), w["conv_post.weight"], w["conv_post.bias"], padding=3)
        mag, ph = torch.clamp(torch.exp(x[:, :9]), max=100), torch.sin(x[:, 9:])
        audio = torch.istft(torch.complex(mag * torch.cos(ph), mag * torch.sin(ph)), n_fft=16, hop_length=4, win_length=16, window=self.window)
        return torch.clamp(audio, -0.99, 0.99)


def test_hift_vocoder_matches_libtorch():
    dims = synth.HIFT_TINY
    w = synth.synth_hift_weights(dims, 999)
    T = 11
    mel, phase, noise = synth.synth_hift_inputs(dims, T, 6)
    orc = S.HiftOracle(w)
    th = TorchHift(w)
    with torch.no_grad():
        tm = torch.from_numpy(mel)[None]
        f0_t = th.f0(tm)
        s_t = th.source(f0_t, torch.from_numpy(phase), torch.from_numpy(noise))
        a_t = th.decode(tm, s_t)[0].numpy()
    f0 = orc.f0_predictor(mel)
    assert np.abs(f0 - f0_t[0].numpy()).max() < 2e-4 * max(1.0, np.abs(f0).max())
    assert (f0 > 10).any() and (f0 <= 10).any()              # voiced and unvoiced frames both occur
    # the source on libtorch's own f0 (the phase accumulates f0 over thousands of samples: a 1e-5 relative difference in f0
    # would dominate the comparison of everything behind it)
    s = orc.source(f0_t[0].numpy(), phase, noise)
    assert s.shape == (T * 480,) and np.abs(s - s_t[0, 0].numpy()).max() < 2e-4
    stft = orc.stft(s_t[0, 0].numpy())
    with torch.no_grad():
        sp = torch.stft(s_t[0], n_fft=16, hop_length=4, win_length=16, window=th.window, return_complex=True)[0]
    assert stft.shape == (18, T * 120 + 1)
    assert np.abs(stft - torch.cat([sp.real, sp.imag], dim=0).numpy()).max() < 1e-5
    audio = orc.decode(mel, s_t[0, 0].numpy())
    assert audio.shape == (T * 480,) and np.isfinite(audio).all()
    assert np.abs(audio - a_t).max() < 2e-4, np.abs(audio - a_t).max()
    assert 0.01 < np.abs(audio).mean() and (np.abs(audio) < 0.99).mean() > 0.5     # neither silent nor pinned at the clip


def test_istft_inverts_stft_of_the_oracle():
    orc = S.HiftOracle(synth.synth_hift_weights(synth.HIFT_TINY, 999))
    x = np.random.default_rng(0).standard_normal(480).astype(np.float32)
    st = orc.stft(x)
    re, im = st[:9], st[9:]
    back = orc.istft(np.hypot(re, im), np.arctan2(im, re))
    assert np.abs(back - x).max() < 1e-4
