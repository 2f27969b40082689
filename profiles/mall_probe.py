"""Does the Infinity Cache (256 MB, memory side) keep freshly WRITTEN lines for a reader that follows?  (The adaptive chain writes
17.2 GB of snapshots and reads them back once: if a chunk that was just written is still on chip when the canceller reads it, a
persistent canceller that follows the analysis bank chunk by chunk saves the read side of that round trip.)  A copy kernel reads
a buffer of `MB` right after a fill kernel wrote it (warm), and after 2 GB of other traffic went through (cold)."""
import sys, os, json
import torch
dev = torch.device("cuda:0")
big = torch.empty(512 * 1024 * 1024, dtype=torch.float32, device=dev)        # 2 GB of other traffic
res = {}
for MB in (32, 64, 128, 192, 256, 512):
    n = MB * 1024 * 1024 // 4
    x = torch.empty(n, dtype=torch.float32, device=dev)
    y = torch.empty(n, dtype=torch.float32, device=dev)
    def timed(cold):
        ts = []
        for _ in range(6):
            x.fill_(1.0)                                    # writer
            if cold:
                big.fill_(2.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            s = x.sum()                                     # reader: n * 4 bytes in
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return min(ts[1:])
    tw, tc = timed(False), timed(True)
    res["%dMB" % MB] = {"warm_read_GBps": MB / 1024 / (tw * 1e-3), "cold_read_GBps": MB / 1024 / (tc * 1e-3)}
print(json.dumps(res))
