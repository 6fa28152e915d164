#!/usr/bin/env python3
"""Host-to-device and device-to-host copy rates of the box (pinned and pageable, one stream and two),
to tell what the link gives from what the host batch path reaches.  Run on the GPU box."""
import time

import torch


def rate(fn, nbytes, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return nbytes * reps / (time.perf_counter() - t0) / 1e9


def main():
    for mb in (8, 32, 256):
        n = mb << 20
        pinned = torch.empty(n, dtype=torch.uint8).pin_memory()
        pageable = torch.empty(n, dtype=torch.uint8)
        pageable.fill_(1)
        dev = torch.empty(n, dtype=torch.uint8, device="cuda")
        print("%4d MiB  H2D pinned %.1f GB/s, pageable %.1f GB/s;  D2H pinned %.1f GB/s, pageable %.1f GB/s"
              % (mb, rate(lambda: dev.copy_(pinned, non_blocking=True), n), rate(lambda: dev.copy_(pageable), n),
                 rate(lambda: pinned.copy_(dev, non_blocking=True), n), rate(lambda: pageable.copy_(dev), n)))
    # two streams, pinned, both directions at once
    n = 256 << 20
    a, b = torch.empty(n, dtype=torch.uint8).pin_memory(), torch.empty(n, dtype=torch.uint8).pin_memory()
    da, db = torch.empty(n, dtype=torch.uint8, device="cuda"), torch.empty(n, dtype=torch.uint8, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def both():
        with torch.cuda.stream(s1):
            da.copy_(a, non_blocking=True)
        with torch.cuda.stream(s2):
            b.copy_(db, non_blocking=True)
    print("256 MiB each way at once: %.1f GB/s per direction" % rate(both, n))
    # host memcpy rate of one thread (the staging copy into pinned memory)
    src = torch.empty(n, dtype=torch.uint8)
    src.fill_(2)
    t0 = time.perf_counter()
    for _ in range(5):
        a.copy_(src)
    print("host memcpy pageable -> pinned, one thread: %.1f GB/s" % (n * 5 / (time.perf_counter() - t0) / 1e9))


if __name__ == "__main__":
    main()
