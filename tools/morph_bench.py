"""Morph-space passes against the HBM roof (N = 1M by default): HIP events around the weights
kernel pair and the displacement kernel, algorithmic bytes = one pass over the matrix."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from facedeform_amd import capi, synth

def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    Ss = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "16,50,100".split(","))]
    dev = torch.device("cuda", 0)
    rest = synth.head_mesh(N)
    d_rest = torch.from_numpy(rest).to(dev)
    g = torch.Generator(device=dev); g.manual_seed(1)
    stream = torch.cuda.Stream(device=dev)
    for S in Ss:
        d_shapes = [d_rest + 0.05 * torch.randn(N, 3, device=dev, generator=g) for _ in range(S)]
        d_P = (d_rest + 0.3 * (d_shapes[0] - d_rest)).contiguous()
        torch.cuda.synchronize()
        m = capi.Morph()
        m.init_dev(N, d_rest.data_ptr(), [t.data_ptr() for t in d_shapes])
        del d_shapes
        rows = 3 * N
        res = {}
        for what in ("weights", "displace"):
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
            for a, b in evs:
                work = d_P.clone()
                m.compute_weights_dev(work.data_ptr(), stream.cuda_stream) if what == "displace" else None
                torch.cuda.synchronize()
                a.record(stream)
                if what == "weights":
                    m.compute_weights_dev(work.data_ptr(), stream.cuda_stream)
                else:
                    m.displace_dev(work.data_ptr(), (-1.0, 1.0), True, 0.5, stream.cuda_stream)
                b.record(stream)
                stream.synchronize()
            ts = sorted(a.elapsed_time(b) for a, b in evs[2:])
            res[what] = ts[len(ts) // 2]
        bw = rows * S * 8 + rows * 4 * 2            # packed QR fp64 + P + rest
        bd = rows * S * 4 + rows * 4 * 3            # fp32 deltas + P in/out + rest
        print(f"N={N} S={S}: QR init {m.last_init_ms:8.1f} ms | weights {res['weights']*1e3:8.1f} us = {bw/res['weights']/1e6:7.0f} GB/s "
              f"({bw/res['weights']/1e6/8000*100:4.1f}% of 8 TB/s) | displace {res['displace']*1e3:8.1f} us = {bd/res['displace']/1e6:7.0f} GB/s "
              f"({bd/res['displace']/1e6/8000*100:4.1f}%)", flush=True)
        m.close()

if __name__ == "__main__":
    main()
