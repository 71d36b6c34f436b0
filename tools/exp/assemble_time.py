import sys, torch
sys.path.insert(0, '.')
import cpi_amd
from cpi_amd import synth
eng = cpi_amd.Engine(device=0)
for W in (1000000, 100000):
    stream, upd, lin, q = synth.make_stream(W, 50, seed=3, device=eng.device, phase=0.4)
    tiles, cnt = eng.assemble_tiles(stream, upd, 51)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(5):
        e0.record()
        eng.assemble_tiles(stream, upd, 51, tiles=tiles, count=cnt)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    gb = (stream.numel() * 8 + tiles.numel() * 8) / 1e9
    print("assemble_tiles W=%d: %.3f ms  (%.2f GB moved -> %.2f TB/s)" % (W, best, gb, gb / best))
    # tile_knots from CSR for comparison
    del tiles
