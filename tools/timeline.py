"""ONE entry point for the per-wave `s_memtime` cycle splits of the conv kernels' main loops (measurement builds only:
tools/build_variant.sh tl "-DYV3_TIMELINE ..."; YV3_MEASURE=1 YV3_LIB=.../libyv3_tl.so).

    python tools/timeline.py --kernel w4 [layer ...]        (environment: BB = batch, WG = workgroup sampled, DT = f32h2 | bf16)

kernels (tools/timelines/<name>.py holds what differs: the descriptor setup that selects the kernel and the layout of its dump):
  pp         eight-wave ping-pong plane kernel            np      its single-phase loop            ps    the K-split ping-pong experiment
  roll       rolling loop (fp16 planes)                   roll_bf16  the bf16 192x256 / 256x256 rolling tiles
  w4         four-wave 192x128 tile, 2 workgroups per CU  wino    Winograd F(2x2,3x3) stage of the plane kernels
  probe      the bare DMA / read / MFMA probe loop        front / front_f32   the fused first-two-layers kernels
The exact-fp32 F(4x4,3x3) stage (round 6) prints its split through tools/wino4_ab.py (TL=1, -DW4F_TIMELINE=<workgroup>)."""
import os
import runpy
import sys

if __name__ == "__main__":
    if len(sys.argv) < 3 or sys.argv[1] != "--kernel":
        sys.exit(__doc__)
    name = sys.argv[2]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "timelines", name + ".py")
    if not os.path.exists(path):
        sys.exit("unknown kernel %r\n\n%s" % (name, __doc__))
    sys.argv = [path] + sys.argv[3:]
    runpy.run_path(path, run_name="__main__")
