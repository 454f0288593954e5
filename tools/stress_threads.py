"""Concurrent context creation / graph capture / readback from several host threads, many rounds (looks for the
HIP error 906 race: a legacy-stream operation in one thread while another thread captures a graph)."""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hopperrender_amd import synth
from hopperrender_amd.calc import OpticalFlowCalcHDR, OpticalFlowCalcSDR, DeviceBuffer
errors = []
def worker(idx, hdr, H, W, seed):
    try:
        sc = synth.Scene(H, W, bool(hdr), seed)
        f = [sc.frame(k) for k in range(3)]
        for rep in range(3):
            c = (OpticalFlowCalcHDR if hdr else OpticalFlowCalcSDR)(H, W, search_radius=5 + (idx + rep) % 10)
            for _ in range(3):
                for x in f: c.updateFrame(x)
                c.calculateOpticalFlow()
                c.readOffsets(); c.readBlurredFlow(1)
                b = DeviceBuffer(f[0].nbytes); b.upload(f[0]); b.download(f[0].dtype, 16); b.free()
            c.close()
    except Exception as e:
        errors.append(repr(e)[:300])
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for r in range(rounds):
    th = [threading.Thread(target=worker, args=(i, i % 2, 90 + 2 * i, 160 + 4 * i, 50 + i + r)) for i in range(6)]
    for t in th: t.start()
    for t in th: t.join()
print("rounds", rounds, "errors", len(errors), errors[:2])
sys.exit(1 if errors else 0)
