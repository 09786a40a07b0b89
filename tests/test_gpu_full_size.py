"""Size-independent properties at BASELINE.json's full sizes (the oracle takes ~10 s of CPU for one such block: it checks a sample of the streams here, all of
them in bench.py's cpu_baseline leg): block-cut invariance of the resynthesis path and of gen.voice, and that no stream depends on its neighbours."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SR = 48000


@pytest.fixture(scope="module")
def env():
    import torch
    import bench
    from cpp_audio_amd import api
    assert api.lib().ra_device_count() > 0
    return torch, bench, api


def test_configs1_full_size_block_cuts_and_stream_independence(env, oracle):
    # configs[1]: 256 streams x 10 s, N 4096, H 1024, P 64.  One 10 s block against the same input cut at 3.7 s (mid-hop): identical events, audio to
    # rounding; stream 200 of the batch equals the same stream run alone with its global seed; three streams against the oracle.
    torch, bench, api = env
    S, n = 256, 10 * SR
    dev = torch.device("cuda", 0)
    x = bench.synth_inputs_torch(S, n, SR, 1234, dev)
    kw = dict(window_size=4096, stride=1024, n_channels=64, analysis_volume=1.0, stereo_spread=0.0, carrier_volume=0.0, seed=1234)
    cfg = api.default_config(**kw)
    out = [torch.empty(S, n, 2, dtype=torch.float32, device=dev) for _ in range(2)]
    with api.Context(cfg, S, 0) as ctx:
        ctx.process_device(x.data_ptr(), 0, n, out[0].data_ptr())
        ctx.synchronize()
        ev_whole = [ctx.events(s) for s in (0, 17, 255)]
        st = ctx.stats()
    assert st.error_flags == 0
    cut = int(3.7 * SR) + 333
    with api.Context(cfg, S, 0) as ctx:
        a, b = x[:, :cut].contiguous(), x[:, cut:].contiguous()
        oa, ob = torch.empty(S, cut, 2, dtype=torch.float32, device=dev), torch.empty(S, n - cut, 2, dtype=torch.float32, device=dev)
        ctx.process_device(a.data_ptr(), 0, cut, oa.data_ptr())
        ctx.synchronize()
        ev_a = [ctx.events(s) for s in (0, 17, 255)]
        ctx.process_device(b.data_ptr(), 0, n - cut, ob.data_ptr())
        ctx.synchronize()
        ev_b = [ctx.events(s) for s in (0, 17, 255)]
        out[1] = torch.cat([oa, ob], dim=1)
    d = (out[0].double() - out[1].double())
    assert float(out[0].double().pow(2).mean().sqrt()) > 1e-2
    assert float(d.abs().max()) < 1e-6 and float(d.pow(2).mean().sqrt()) < 1e-8
    for w, ea, eb in zip(ev_whole, ev_a, ev_b):
        both = np.concatenate([ea, eb])
        assert len(w) == len(both) and len(w) > 1000
        for f in ("sample", "type", "channel", "noteid", "frequency", "velocity"):
            assert np.array_equal(w[f], both[f]), f
    # a stream of the batch is the stream alone (per-stream seeds = seed + global index)
    s = 200
    with api.Context(api.default_config(**dict(kw, seed=1234 + s)), 1, 0) as ctx:
        alone = torch.empty(1, n, 2, dtype=torch.float32, device=dev)
        ctx.process_device(x[s:s + 1].contiguous().data_ptr(), 0, n, alone.data_ptr())
        ctx.synchronize()
    assert torch.equal(alone[0], out[0][s])
    # and the oracle on a sample of the streams
    hx = x[[3, 100, 255]].cpu().numpy()
    for i, s in enumerate((3, 100, 255)):
        ref, _, _ = oracle.run_offline(oracle.default_config(**dict(kw, seed=1234 + s)), hx[i])
        got = out[0][s].cpu().numpy()
        assert np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2)) <= 1e-7, s


@pytest.mark.parametrize("which", ["sweep", "birds"])
def test_voice_full_size_block_cuts(env, oracle, which):
    # the bench's gen.voice workloads (256 synths x 16 channels x 10 s): one block against three ragged ones (another cut packs the ramps into other
    # slots: the sums associate differently), no dropped note, and two synths against the oracle
    torch, bench, api = env
    S, n = 256, 10 * SR
    dev = torch.device("cuda", 0)
    cfg, rows, _ = bench.voice_workload(api, which, S, 10.0, 1)
    outs = []
    for cuts in ((n,), (100001, 7, n - 100008)):
        with api.Voice(cfg, S, 0) as v:
            v.events(rows)
            parts = []
            for m in cuts:
                o = torch.empty(S, m, 2, dtype=torch.float32, device=dev)
                v.process_device(m, o.data_ptr())
                parts.append(o)
            assert v.dropped_note_ons() == 0
        outs.append(torch.cat(parts, dim=1))
    d = outs[0].double() - outs[1].double()
    assert float(outs[0].double().pow(2).mean().sqrt()) > 1e-3
    assert float(d.abs().max()) < 1e-6
    for s in (0, 131):
        ref, dropped = oracle.voice_run(oracle.voice_params_from(cfg, cfg.seed + s), [(r[0], r[2], r[3], r[4], r[5]) for r in rows if r[1] == s], n)
        got = outs[0][s].cpu().numpy()
        assert dropped == 0 and np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2)) <= 6e-8, s


def test_note_deduction_of_a_batch_at_full_size(env):
    # driver (A) over a folder: 64 signals of 10 s (main.resynth's half window 400 / stride 400) through ra_deduce_notes_batch — one context, two blocks —
    # against single calls for a sample of them; notes come out ordered by closing frame, every note spans at least one frame and lies inside the signal
    torch, bench, api = env
    sr, S = 44100, 64
    x = bench.synth_inputs_torch(S, 10 * sr, sr, 99, torch.device("cuda", 0)).cpu().numpy()
    got = api.deduce_notes_batch(x, sr, 400, 400)
    assert len(got) == S
    n_frames = 10 * sr // 400 + 1
    for s in range(S):
        g = got[s]
        assert len(g) > 10
        assert np.all(g["start_frame"] <= g["end_frame"]) and g["start_frame"].min() >= 0 and g["end_frame"].max() <= n_frames
        assert np.all(np.diff(g["end_frame"]) >= 0)
        assert np.all(g["frequency"] > 0) and np.all(np.isfinite(g["amplitude_db"]))
    for s in (0, 31, 63):
        assert np.array_equal(got[s], api.deduce_notes(x[s], sr, 400, 400))
