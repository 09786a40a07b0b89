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


@pytest.mark.parametrize("which", ["sweep", "birds", "wind"])
def test_voice_full_size_block_cuts(env, oracle, which):
    # the bench's gen.voice workloads (256 synths x 16 channels x 10 s; WIND: four held notes per synth through band-passes of order 89): one block against
    # three ragged ones (another cut packs the ramps into other slots: the sums associate differently), no dropped note, and two synths against the oracle
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
    assert float(outs[0].double().pow(2).mean().sqrt()) > (1e-4 if which == "wind" else 1e-3)
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


@pytest.mark.parametrize("kind", [0, 1])
def test_configs2_full_size_vocoder(env, oracle, kind):
    # BASELINE configs[2] at its full size — 256 streams x 10 s, 32 bands — for both vocoders: the FFT vocoder with the carrier from a file (the offline
    # job's way), the band-pass bank fed by the carrier synth (the bench's voc_fft / voc_bank workloads).  With the resynthesis silent the output IS the
    # vocoder's mix: one 10 s block against the same input cut mid-frame is equal bit for bit; eight streams against the oracle.
    torch, bench, api = env
    S, n = 256, 10 * SR
    dev = torch.device("cuda", 0)
    desc, kw, S2, carrier = bench.workloads()["voc_fft" if kind == 0 else "voc_bank"]
    assert S2 == S
    x = bench.synth_inputs_torch(S, n, SR, 4321, dev)
    car = bench.carrier_saw_torch(S, n, SR, dev) if carrier == "wav" else None
    rows = bench.carrier_chords(S, 1, n) if carrier == "synth" else None
    cfg = api.default_config(sample_rate=SR, **kw)
    outs = []
    for cuts in ((n,), (int(3.7 * SR) + 333, 1, n - int(3.7 * SR) - 334)):
        parts, t = [], 0
        with api.Context(cfg, S, 0) as ctx:
            if rows:
                ctx.carrier_events(rows)
            for m in cuts:
                o = torch.empty(S, m, 2, dtype=torch.float32, device=dev)
                a = x[:, t:t + m].contiguous()
                c = car[:, t:t + m].contiguous() if car is not None else None
                ctx.process_device(a.data_ptr(), c.data_ptr() if c is not None else 0, m, o.data_ptr())
                ctx.synchronize()
                parts.append(o)
                t += m
            st = ctx.stats()
            assert st.error_flags == 0 and st.carrier_dropped_note_on == 0
        outs.append(torch.cat(parts, dim=1))
    assert float(outs[0].double().pow(2).mean().sqrt()) > 1e-3
    if kind == 0:
        assert torch.equal(outs[0], outs[1]), "the FFT vocoder's output does not depend on the block cut"
    else:  # the time-parallel bank's scan tiles start where the block does: equal to rounding (bit for bit with reference_order = 1)
        dd = outs[0].double() - outs[1].double()
        assert float(dd.abs().max()) <= 1e-6 and float(dd.pow(2).mean().sqrt()) <= 1e-9
    sample = (0, 1, 77, 128, 200, 254, 255, 31)
    hx = x[list(sample)].cpu().numpy()
    hc = car[list(sample)].cpu().numpy() if car is not None else None
    okw = {k: v for k, v in kw.items()}
    for i, s in enumerate(sample):
        ocfg = oracle.default_config(sample_rate=SR, **dict(okw, seed=kw["seed"] + s))
        if rows:
            ev = oracle.carrier_events([(r[0], r[2], r[3], r[4], r[5], r[6]) for r in rows if r[1] == s])
            ref, _, _, _ = oracle.run_offline_ex(ocfg, hx[i], None, ev)
        else:
            ref, _, _ = oracle.run_offline(ocfg, hx[i], hc[i])
        got = outs[0][s].cpu().numpy()
        assert np.sqrt(np.mean(ref.astype(np.float64) ** 2)) > 1e-3
        assert np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2)) <= 1e-5, s


def test_configs3_full_size(env, oracle):
    # BASELINE configs[3] at its full size: 1000 stereo = 2000 mono streams x 10 s, N 8192, H 2048, P 128, autotune (major scale on C).  One block against
    # the input cut mid-hop (identical events on a sample of the streams, audio to rounding), stream 1500 of the batch = the same stream alone with its
    # global seed (bit for bit), and sixteen streams against the oracle: with autotune every decision sits on a tie, so this only holds because the
    # decision path is bit-reproducible.
    torch, bench, api = env
    desc, kw, S, _ = bench.workloads()["configs3"]
    n = 10 * SR
    assert S == 2000 and kw["window_size"] == 8192 and kw["n_channels"] == 128 and kw["use_autotune"] == 1
    dev = torch.device("cuda", 0)
    x = bench.synth_inputs_torch(S, n, SR, 4321, dev)
    cfg = api.default_config(sample_rate=SR, **kw)
    watch = (0, 999, 1500, 1999)
    whole = torch.empty(S, n, 2, dtype=torch.float32, device=dev)
    with api.Context(cfg, S, 0) as ctx:
        ctx.process_device(x.data_ptr(), 0, n, whole.data_ptr())
        ctx.synchronize()
        ev_whole = [ctx.events(s) for s in watch]
        assert ctx.stats().error_flags == 0
    cut = int(4.3 * SR) + 1111
    with api.Context(cfg, S, 0) as ctx:
        a, b = x[:, :cut].contiguous(), x[:, cut:].contiguous()
        oa, ob = torch.empty(S, cut, 2, dtype=torch.float32, device=dev), torch.empty(S, n - cut, 2, dtype=torch.float32, device=dev)
        ctx.process_device(a.data_ptr(), 0, cut, oa.data_ptr())
        ctx.synchronize()
        ev_a = [ctx.events(s) for s in watch]
        ctx.process_device(b.data_ptr(), 0, n - cut, ob.data_ptr())
        ctx.synchronize()
        ev_b = [ctx.events(s) for s in watch]
        assert ctx.stats().error_flags == 0
    d = torch.cat([oa, ob], dim=1).double() - whole.double()
    assert float(whole.double().pow(2).mean().sqrt()) > 1e-2
    assert float(d.abs().max()) < 1e-6 and float(d.pow(2).mean().sqrt()) < 1e-8
    del d, oa, ob, a, b
    for w, ea, eb in zip(ev_whole, ev_a, ev_b):
        both = np.concatenate([ea, eb])
        assert len(w) == len(both) and len(w) > 300
        for f in ("sample", "type", "channel", "noteid", "frequency", "velocity"):
            assert np.array_equal(w[f], both[f]), f
    s = 1500
    with api.Context(api.default_config(sample_rate=SR, **dict(kw, seed=kw["seed"] + s)), 1, 0) as ctx:
        alone = torch.empty(1, n, 2, dtype=torch.float32, device=dev)
        ctx.process_device(x[s:s + 1].contiguous().data_ptr(), 0, n, alone.data_ptr())
        ctx.synchronize()
    assert torch.equal(alone[0], whole[s])
    sample = list(range(0, 2000, 133)) + [1999]
    hx = x[sample].cpu().numpy()
    for i, s in enumerate(sample):
        ref, _, _ = oracle.run_offline(oracle.default_config(sample_rate=SR, **dict(kw, seed=kw["seed"] + s)), hx[i])
        got = whole[s].cpu().numpy()
        assert np.sqrt(np.mean(ref.astype(np.float64) ** 2)) > 1e-3
        assert np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2)) <= 1e-5, s
