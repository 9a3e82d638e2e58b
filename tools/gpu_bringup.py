"""Bring-up diagnostics on the GPU box (prints numbers instead of asserting)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle_cpu as oc
from yolo_v3_amd import synth, detect, postprocessing, Detector, _ffi
from tests.helpers import load_sw1_net, rel_err

torch.cuda.set_device(0)
stream = synth.weight_stream()
net = load_sw1_net(stream).cuda()
g = np.load("tests/golden/e2e.npz")
which = sys.argv[1:] or ["e2e", "decisions", "graph"]

if "e2e" in which:
    for name in ["dog416", "u416"]:
        if name == "dog416":
            x = torch.from_numpy(g["dog_u8"].astype(np.float32) / np.float32(255.0)).permute(2, 0, 1).unsqueeze(0).contiguous()
        else:
            x = torch.from_numpy(synth.images(2, 416, int(g[name + "_seed"][0])))
        with torch.no_grad():
            dets = net.forward_cat(x.cuda())
        rows = g[name + "_rows"]
        e = rel_err(dets[:, rows].cpu(), g[name + "_dets_rows"])
        print(name, "det err max", float(e.max()), "mean", float(e.mean()))
        sc = dets[..., 5:] * dets[..., 4:5]
        mx, arg = sc.max(-1)
        cand = torch.cat(((mx > 0.5).nonzero(), arg[mx > 0.5].unsqueeze(1)), 1).cpu().numpy().astype(np.int32)
        gc = g[name + "_cand"]
        print(name, "cand got", cand.shape, "golden", gc.shape, "equal", np.array_equal(cand, gc))
        if not np.array_equal(cand, gc):
            s1, s2 = set(map(tuple, cand.tolist())), set(map(tuple, gc.tolist()))
            print("  only got", sorted(s1 - s2)[:10], "only golden", sorted(s2 - s1)[:10])
        res = postprocessing(dets, 80, 0.5, 0.4)
        ref = oc.postprocess(dets.cpu(), 80, 0.5, 0.4)
        for i, (r, e2) in enumerate(zip(res, ref)):
            gold = g["%s_boxes%d" % (name, i)]
            print(name, i, "gpu", tuple(r.shape), "oracle-on-gpu-dets", tuple(e2.shape), "golden", gold.shape,
                  "gpu==oracle", r.shape == e2.shape and bool(torch.equal(r, e2)))
            if r.shape == e2.shape and not torch.equal(r, e2):
                bad = (r != e2).any(1).nonzero().flatten()[:5]
                print("  first diffs rows", bad.tolist()); print(r[bad]); print(e2[bad])
            elif r.shape != e2.shape:
                print("  gpu classes", r[:, 6].tolist()[:40]); print("  ora classes", e2[:, 6].tolist()[:40])

if "decisions" in which:
    x = torch.from_numpy(synth.images(4, 416, 4242)).cuda()
    with torch.no_grad():
        dets = net.forward_cat(x)
    for (ct, nt, ev) in [(0.5, 0.4, False), (0.3, 0.45, True)]:
        ref = oc.postprocess(dets.cpu(), 80, ct, nt, ev, True)
        res = postprocessing(dets, 80, ct, nt, ev, True)
        res2 = detect(net, x, 80, ct, nt, ev)
        for i in range(4):
            print("decisions", ev, i, tuple(res[i].shape), tuple(res2[i].shape), tuple(ref[i].shape),
                  res[i].shape == ref[i].shape and bool(torch.equal(res[i], ref[i])),
                  res2[i].shape == ref[i].shape and bool(torch.equal(res2[i], ref[i])))

if "graph32" in which:
    base = synth.images(8, 416, 99)
    x = torch.from_numpy(base[[0, 1, 2, 3, 4, 5, 6, 7] * 4]).cuda()
    d0 = Detector(net, 32, 416, 416)
    r0 = d0(x); print("eager32 ok"); sys.stdout.flush()
    gd = Detector(net, 32, 416, 416, graph=True)
    for it in range(3):
        b, c = gd.run_device(x); print("run_device", it); sys.stdout.flush()
        torch.cuda.synchronize(); print("synced", it); sys.stdout.flush()
        h = c.cpu(); print("counts", h[:4].tolist(), h[32:36].tolist()); sys.stdout.flush()
        r = gd.to_list(b, h); print("list ok", all(torch.equal(a, e) for a, e in zip(r, r0))); sys.stdout.flush()

pieces = [w.split(":")[1] for w in which if w.startswith("piece:")]
if pieces:
    B = int(os.environ.get("BB", "4"))
    x = torch.from_numpy(synth.images(B, 416, 99)).cuda()
    d = Detector(net, B, 416, 416)
    r0 = d(x)
    print("eager ok", [tuple(r.shape) for r in r0][:4]); sys.stdout.flush()
    for piece in pieces:
        st = torch.empty_like(x); st.copy_(x)
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            d._enqueue(st)
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        print("capturing", piece); sys.stdout.flush()
        with torch.cuda.graph(gr):
            if piece == "convs": d.engine.run_convs(d.plan, st, d.dets)
            elif piece == "decode": d.engine.run_decode(d.plan, d.dets)
            elif piece == "filter": d.lane_pp[0].filter(d.dets, 0.5, False, True)
            elif piece == "nms": d.lane_pp[0].nms(d.dets, 0.4, True, d.max_cand, d.cap)
            else: d._enqueue(st)
        for it in range(4):
            gr.replay(); torch.cuda.synchronize()
            print("replayed", piece, it, d.counts.cpu()[:4].tolist()); sys.stdout.flush()
