#!/usr/bin/env python3
"""Round 6 probe: does the H2D rate out of a pinned buffer depend on the NUMA node its pages live on (the host path's "6 ms mode":
3.1-3.5 ms of a call waiting for the device although the copy threads were done after 2.2)?  Pins this thread to the CPUs of each node in
turn, allocates + first-touches a pinned buffer there, and times H2D / D2H copies of 107 MB (the int8 wire bytes of a 4096-codeword call)
in 8 chunks on one stream, as the library does.  Prints the GPU's own NUMA node as the kernel reports it."""
import glob, json, os, time
import numpy as np, torch


def node_cpus():
    out = {}
    for p in sorted(glob.glob("/sys/devices/system/node/node*/cpulist")):
        n = int(p.split("node")[-1].split("/")[0])
        cpus = set()
        for part in open(p).read().strip().split(","):
            if "-" in part:
                a, b = part.split("-"); cpus |= set(range(int(a), int(b) + 1))
            elif part:
                cpus.add(int(part))
        out[n] = cpus
    return out


def gpu_nodes():
    res = {}
    for p in glob.glob("/sys/class/drm/card*/device/numa_node"):
        try:
            res[p.split("/")[4]] = int(open(p).read())
        except Exception:  # noqa: BLE001
            pass
    return res


def main():
    nodes = node_cpus()
    allowed = os.sched_getaffinity(0)
    print(json.dumps({"nodes": {n: len(c) for n, c in nodes.items()}, "gpu_numa_nodes": gpu_nodes(), "allowed_cpus": len(allowed)}), flush=True)
    dev = torch.device("cuda", 0)
    nbytes, chunks = 107 * 1000 * 1000, 8
    d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    for rep in range(2):
        for n, cpus in nodes.items():
            use = cpus & allowed
            if not use:
                continue
            os.sched_setaffinity(0, use)
            time.sleep(0.01)
            h = torch.empty(nbytes, dtype=torch.uint8).pin_memory()  # hipHostMalloc + first touch on this node's CPUs
            h.fill_(1)
            res = {}
            for name, src, dst in (("h2d", h, d), ("d2h", d, h)):
                ts = []
                for _ in range(7):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    step = nbytes // chunks
                    for k in range(chunks):
                        dst[k * step:(k + 1) * step].copy_(src[k * step:(k + 1) * step], non_blocking=True)
                    e1.record(); e1.synchronize()
                    ts.append(e0.elapsed_time(e1))
                ts.sort()
                res[name] = {"ms_median": ts[3], "GB_s": nbytes / ts[3] / 1e6}
            print(json.dumps({"pinned_buffer_touched_on_node": n, **res}), flush=True)
            del h
    os.sched_setaffinity(0, allowed)


if __name__ == "__main__":
    main()
