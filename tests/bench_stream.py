"""Streaming-read rate of one MI355X with different access shapes (calibration for the cross-attention roofline; not collected by
pytest).  python tests/bench_stream.py [GiB]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from gram_amd import _lib
    lib = _lib.load()
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
    n = int(gib * 2 ** 30)
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    buf.fill_(3)
    sink = torch.zeros(4, dtype=torch.int32, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for variant in ((5, 6) if gib > 16 else (0, 1, 2, 3, 4, 5, 6, 7)):
        for wgs in (1024, 2048, 4096):
            for _ in range(2):
                _lib.check(lib.gram_debug_stream_read_variant(buf.data_ptr(), n, sink.data_ptr(), variant, wgs, st), "probe")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                _lib.check(lib.gram_debug_stream_read_variant(buf.data_ptr(), n, sink.data_ptr(), variant, wgs, st), "probe")
            e1.record()
            torch.cuda.synchronize()
            out.append({"variant": variant, "wgs": wgs, "TBps": round(n * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e12, 3)})
            print(json.dumps(out[-1]), flush=True)


if __name__ == "__main__":
    main()
