"""developer tool: pack the synthetic weights with two builds of the library (CBGX_LIBRARY) and report which regions of the
packed blob differ.  Usage on the GPU box: python scripts/dev_pack_diff.py ab_libs/old.so cbgbench_amd/lib/libcbgx.so"""
import os, subprocess, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 3 and sys.argv[1] == "--dump":
    sys.path.insert(0, ROOT)
    import torch, cbgbench_amd as C
    from cbgbench_amd import synthetic_weights
    m = C.get_model(C.default_targetdiff_config(13)).eval()
    synthetic_weights.fill_(m, seed=0)
    m = m.to("cuda:0")
    p = m.denoiser.packed_weights(torch.device("cuda:0"))
    torch.cuda.synchronize()
    np.save(sys.argv[2], p.cpu().numpy())
    sys.exit(0)
blobs = []
for k, lib in enumerate(sys.argv[1:3]):
    out = f"/tmp/blob{k}.npy"
    subprocess.run([sys.executable, __file__, "--dump", out], check=True, env=dict(os.environ, CBGX_LIBRARY=os.path.abspath(lib)))
    blobs.append(np.load(out))
a, b = blobs
print("sizes", a.size, b.size)
H, HEADS, G, NT, GH, KV_IN, PROW = 128, 16, 20, 4, 160, 340, 640
off, reg = 0, []
def R(name, n):
    global off
    reg.append((name, off, n)); off += n
GATE = GH * G + 4 * GH + 4 + (GH // 16) * 5 * 64 + 4 * GH
FRAG = NT * 8 * 320
names = [("A_WN", H * PROW), ("A_BN", PROW), ("A_WT", NT * 2 * H), ("A_WR", NT * G * 2 * H), ("LN6", 6 * H), ("A_WQ1T", H * H), ("A_BQ1", H),
         ("A_WBK", H * H), ("A_WBV", H * H), ("A_BBV", H), ("IMG_FRAG_K", FRAG), ("IMG_FRAG_V", FRAG), ("IMG_WT", NT * 2 * H), ("IMG_LN", 4 * H),
         ("IMG_WBV", H * H), ("A_NPROJ_FRAG", H * PROW), ("A_WQ1_FRAG", H * H), ("A_WBK_FRAG", H * H), ("A_WAKC", H * KV_IN), ("A_BAKC", H),
         ("A_WAVC", H * KV_IN), ("A_BAVC", H), ("A_BN2", 2 * PROW), ("A_WBKT", H * H), ("A_WQ1O", H * H), ("A_WRT", NT * 2 * H * 32),
         ("A_WRC", NT * G * 2 * H), ("A_FRAGV_EM", FRAG), ("A_RBF_SC", 8), ("A_NPROJ_CINV", PROW), ("A_WQ1_CINV", H)]
att = sum(n for _, n in names)
for l in range(9):
    for blk in ("x2h", "h2x"):
        base = GATE + (2 * l + (blk == "h2x")) * att
        o = base
        for nm, n in names:
            d = np.abs(a[o:o + n] - b[o:o + n])
            bad = int((d > 0).sum()) + int(np.isnan(d).sum())
            if bad and l < 1 and nm.endswith("CINV"):
                idx = np.nonzero(d > 0)[0][:24]
                print("   cols", idx.tolist(), "log2 old", np.log2(a[o + idx]).tolist(), "log2 new", np.log2(b[o + idx]).tolist())
            if bad and l < 2:
                print(f"layer {l} {blk} {nm}: {bad}/{n} differ, max {np.nanmax(d):.3e}; a {a[o:o+4]} b {b[o:o+4]}")
            o += n
print("gate differs:", int((a[:GATE] != b[:GATE]).sum()), " tail differs:", int((a[GATE + 18 * att:] != b[GATE + 18 * att:]).sum()))
