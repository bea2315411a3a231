"""Do two independent CConv launches of a layer group overlap when they are enqueued on two HIP streams?  (Every splat kernel is one
workgroup per CU with most of the LDS: only the tail of one launch can meet the head of the next.)
    python tools/overlap_test.py        -> ms for L3 ; L4 ; L2 back to back on one stream and on three streams"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmcf_amd import ops
from dmcf_amd.utils.tools.losses import grid_pos
from tools import scenes


def main():
    dev = torch.device("cuda:0")
    sc = scenes.box_scene(100)
    s0 = torch.from_numpy(np.concatenate([sc["pos"], sc["box"]])).to(dev)
    s1 = grid_pos(s0, np.float32([0.05] * 3), centralize=True)
    s2 = grid_pos(s0, np.float32([0.1] * 3), centralize=True)
    g = torch.Generator(device=dev).manual_seed(0)
    fns = []
    for out, R, cin, cout in ((s1, 0.2, 24, 8), (s2, 0.4, 24, 4), (s0, 0.1, 24, 16)):
        nns = ops.fixed_radius_search(s0, out, R, return_distances=False)
        feat = torch.rand(s0.shape[0], cin, device=dev, generator=g)
        W = torch.rand(4, 4, 4, cin, cout, device=dev, generator=g) - 0.5
        res = torch.empty(out.shape[0], cout, device=dev)
        fns.append(lambda W=W, out=out, R=R, feat=feat, nns=nns, res=res: ops.cconv_forward(
            W, out, 2 * R, s0, feat, nns.neighbors_index, nns.neighbors_row_splits, window="poly6", out=res,
            row_length_hint=2 if R > 0.1 else 1))
    for f in fns:
        f()
    torch.cuda.synchronize()

    def run(streams):
        ts = []
        for _ in range(5):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record()
            if streams is None:
                for f in fns:
                    f()
            else:
                cur = torch.cuda.current_stream()
                for f, st in zip(fns, streams):
                    st.wait_stream(cur)
                    with torch.cuda.stream(st):
                        f()
                for st in streams:
                    cur.wait_stream(st)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return float(np.median(ts))

    print(f"one stream: {run(None):.2f} ms;  three streams: {run([torch.cuda.Stream() for _ in fns]):.2f} ms")


if __name__ == "__main__":
    main()
