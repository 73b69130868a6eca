#!/usr/bin/env python3
"""What the chip does under the q8gemm 4096^3 kernel: shader clock and socket power sampled from sysfs (hwmon) while one
variant runs back to back for a few seconds.   python tools/gemm_power.py [--variants 15,20] [--seconds 3]
Prints one JSON line per variant: kernel, us per launch, TOP/s, and min / median / max of every sensor that exists."""
import argparse, glob, json, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sensors():
    out = {}
    for pat, key in (("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "power_avg_uW"),
                     ("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input", "power_in_uW"),
                     ("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input", "sclk_Hz"),
                     ("/sys/class/drm/card*/device/hwmon/hwmon*/temp1_input", "temp_mC")):
        for path in sorted(glob.glob(pat))[:1]:
            out[key] = path
    return out


def read(path):
    try:
        return float(open(path).read().split()[0])
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="15,20")
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--size", type=int, default=4096)
    args = ap.parse_args()
    import torch
    import qnnpack_amd
    lib = qnnpack_amd.load()
    torch.cuda.set_device(0); torch.zeros(1, device="cuda")
    lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
    M = N = K = args.size
    rng = np.random.default_rng(0x51A0 + 2)
    w = rng.integers(0, 256, size=(N, K), dtype=np.uint8)
    bias = rng.integers(-10000, 10001, size=N, dtype=np.int32)
    gen = torch.Generator(device="cuda"); gen.manual_seed(0x51A0)
    a = torch.randint(0, 256, (M * K,), dtype=torch.uint8, device="cuda", generator=gen)
    sens = sensors()
    print(json.dumps({"sensors": sens}), flush=True)
    for v in [int(x) for x in args.variants.split(",")]:
        lib.set_option("gemm_kernel", v)
        op = lib.create_fully_connected_nc_q8(K, N, 127, 0.75, 127, 1.0, w, bias, 127, 1.0, 1, 254)
        out = torch.empty(M * N, dtype=torch.uint8, device="cuda")
        lib.setup_fully_connected_nc_q8(op, M, a, K, out, N)
        lib.run_operator(op)
        lib.set_option("gemm_kernel", 0)
        lib.set_async(True)
        lib.graph_begin()
        for _ in range(64):
            lib.run_operator(op)
        g = lib.graph_end()
        lib.graph_time(g, 2, 20)
        samples = {k: [] for k in sens}
        stop = threading.Event()

        def poll():
            while not stop.is_set():
                for k, path in sens.items():
                    x = read(path)
                    if x is not None:
                        samples[k].append(x)
                time.sleep(0.02)
        th = threading.Thread(target=poll); th.start()
        t_end = time.time() + args.seconds
        times = []
        while time.time() < t_end:
            times.append(lib.graph_time(g, 0, 40) / 64.0)
        stop.set(); th.join()
        lib.set_async(False)
        med = sorted(times)[len(times) // 2]
        row = {"gemm_kernel": v, "kernel": lib.operator_kernel(op), "us": round(med * 1e3, 2),
               "tops": round(2.0 * M * N * K / (med * 1e-3) / 1e12, 1)}
        for k, xs in samples.items():
            if xs:
                xs = sorted(xs)
                row[k] = {"min": xs[0], "median": xs[len(xs) // 2], "max": xs[-1], "n": len(xs)}
        print(json.dumps(row), flush=True)
        lib.graph_destroy(g); lib.delete_operator(op)


if __name__ == "__main__":
    main()
