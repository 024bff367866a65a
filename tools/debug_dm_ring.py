import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import dsabeamformer_amd as bfm
from dsabeamformer_amd import _lib, api
import oracle as orc
hip = _lib._preload_hip_runtime()
rng = np.random.default_rng(77)
n_t, n_f, n_b, n_dm, max_rows = 210, 48, 256, 40, 32
delays = np.ascontiguousarray((np.arange(n_dm)[:, None] * np.linspace(45 / max(n_dm - 1, 1), 0.0, n_f)[None, :]).astype(np.int32))
D = int(delays.max())
series = (rng.random((n_t, n_f, n_b), dtype=np.float32) * 1e3).astype(np.float32)
want = orc.dedisperse_dm(series, delays, n_t - D)
bf = bfm.Beamformer(bfm.debug_config(n_beams=n_b, n_freq=n_f))
d_series = torch.from_numpy(series).cuda()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
row_bytes = n_f * n_b * 4
import itertools
ERR = int(os.environ.get("ERRCALLS", "0"))
for wide, feed, ring in [(1, "copy", 1), (0, "copy", 0), (1, "reserve", 1), (0, "mixed", 1), (1, "mixed", 1), (0, "reserve", 1), (1, "mixed", 0), (0, "reserve", 0), (0, "mixed", 1)]:
  for _ in (0,):
    for __ in (0,):
        bf.set_switch("dm_wide", wide); bf.set_switch("dm_ring", ring)
        dm = api.DmStream(bf, delays, n_f, max_rows)
        host = torch.full((n_dm * max_rows * n_b,), float("nan"), dtype=torch.float32).pin_memory()
        parts, at, pushed, k = [], 0, 0, 0
        sizes = [max_rows, 1, 7, max_rows, 3, 19, 2, max_rows, max_rows - 1, 11]
        log = []
        while pushed < n_t:
            n = min(sizes[k % len(sizes)], n_t - pushed)
            st = streams[k % 2]
            src = d_series.data_ptr() + pushed * row_bytes
            mode = "c"
            if feed == "reserve" or (feed == "mixed" and k % 3):
                dst = dm.reserve(n, st.cuda_stream)
                if ERR:
                    try:
                        dm.reserve(n, st.cuda_stream)
                    except Exception as e:
                        pass
                    try:
                        dm.push(src, n, host, st.cuda_stream)
                    except Exception as e:
                        pass
                assert hip.hipMemcpyAsync(C.c_void_p(dst), C.c_void_p(src), C.c_size_t(n * row_bytes), 3, C.c_void_p(st.cuda_stream)) == 0
                src = dst; mode = "r"
            first, n_out = dm.push(src, n, host, st.cuda_stream)
            st.synchronize()
            if n_out:
                parts.append(host[:n_dm * n_out * n_b].numpy().reshape(n_dm, n_out, n_b).copy())
                w = want[:, at:at+n_out]
                bad = (parts[-1] != w)
                log.append("%s%d:%s" % (mode, n, "ok" if not bad.any() else "BAD%d(t %s)" % (bad.sum(), sorted(set(np.nonzero(bad)[1]))[:6])))
            at += n_out; pushed += n; k += 1
        got = np.concatenate(parts, axis=1)
        print("wide", wide, feed, "ring", ring, "equal", np.array_equal(got, want), " ".join(log), flush=True)
        dm.close()
