"""ra_process_host_pipelined (chunked copy-in / block / copy-out on three queues) and the page-locked host buffers of ra_host_alloc: the chunks are
ordinary consecutive blocks, so the result equals ra_process_host's up to the rounding of a block cut — bit for bit in reference order — from
pageable and from page-locked memory, with and without a carrier."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from cpp_audio_amd import api
    assert api.lib().ra_device_count() > 0
    return api


def rms(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


@pytest.mark.parametrize("ro", [0, 1])
def test_pipelined_host_blocks_equal_one_host_block(api, ro):
    from cpp_audio_amd import signals
    S, n = 6, 100000
    x = signals.make_streams(S, n, base_seed=77)
    kw = dict(window_size=2048, stride=512, n_channels=32, analysis_volume=1.0, stereo_spread=0.0, carrier_volume=0.0, seed=5, reference_order=ro)
    with api.Context(api.default_config(**kw), S, 0) as ctx:
        whole = ctx.process_host(x)
        st = ctx.stats()
    assert st.error_flags == 0 and np.sqrt(np.mean(whole.astype(np.float64) ** 2)) > 1e-3
    for chunk in (7001, 48000, 250000):   # ragged chunks, about a second, more than the block
        with api.Context(api.default_config(**kw), S, 0) as ctx:
            got = ctx.process_host(x, chunk_frames=chunk)
            # a second call continues the streams: two pipelined calls = one block of both
            got2 = ctx.process_host(x[:, :30000], chunk_frames=chunk)
        with api.PinnedArray((S, n), np.float32) as pi, api.PinnedArray((S, n, 2), np.float32) as po, api.Context(api.default_config(**kw), S, 0) as ctx:
            pi.array[...] = x
            ctx.process_host(pi.array, out=po.array, chunk_frames=chunk)
            pinned = po.array.copy()
        assert np.array_equal(got, pinned), "pageable and page-locked buffers: the same bytes through the same blocks"
        if ro == 1:
            assert np.array_equal(got, whole), chunk
        else:
            assert rms(got, whole) <= 1e-8 and np.max(np.abs(got.astype(np.float64) - whole)) <= 1e-6, chunk
        assert got2.shape == (S, 30000, 2) and np.all(np.isfinite(got2))


def test_pipelined_host_blocks_with_a_carrier(api):
    from cpp_audio_amd import signals
    S, n = 3, 60000
    x = signals.make_streams(S, n, kind="speech", base_seed=3)
    car = np.stack([signals.carrier_stream(n, freq=90.0 + 20 * s) for s in range(S)])
    kw = dict(analysis_volume=0.0, voice_volume=0.2, carrier_volume=0.1, vocoder_volume=2.0, vocoder_count_bands=16, vocoder_min_freq=100.0, vocoder_max_freq=12000.0,
              stereo_spread=0.0, seed=9)
    with api.Context(api.default_config(**kw), S, 0) as ctx:
        whole = ctx.process_host(x, car)
    with api.Context(api.default_config(**kw), S, 0) as ctx:
        got = ctx.process_host(x, car, chunk_frames=9999)
        assert ctx.stats().error_flags == 0
    assert np.sqrt(np.mean(whole.astype(np.float64) ** 2)) > 1e-3
    assert np.array_equal(got, whole), "the FFT vocoder's mix does not depend on the block cut"


def test_pinned_array_lifecycle(api):
    with api.PinnedArray((4, 1000), np.float32) as a:
        a.array[...] = 1.5
        assert a.array.sum() == 6000.0
    assert a.array is None
