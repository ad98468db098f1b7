#!/usr/bin/env python3
"""Reference-style experiment harness for the dpu_snappy CLI.

Counterpart of the reference's snappy/scripts/asplos21/run_tests.py (which rebuilds the tool per
(NR_DPUS, NR_TASKLETS) and scrapes its stdout): here the GPU count is a runtime flag (-g), and the
same stdout lines are scraped.  Written into --out:

  in the reference's own layouts (so its chart_breakdown.py / chart_dpu_speedup.py conventions apply unchanged; the
  device-count column keeps the reference's name `dpus` and holds the number of GPUs):
    <file>_compression_breakdown.csv, <file>_decompression_breakdown.csv
                    prepare, alloc, load, copy_in, run, copy_out, free, dpus        (run_tests.py:134, :146)
    compression_speedup_dpu.csv, decompression_speedup_dpu.csv
                    version, time, dpus        one `host,1,0` row, then <file>, host time / (GPU run + overheads), count
                                               (run_tests.py:86-101, :104-119)
  and, as documented extras:
    speedup.csv     file, bytes, direction, gpus, host_s, gpu_kernel_s, gpu_total_s, speedup_kernel, speedup_total
                    (host_s = "Host time" of the CPU mode, as run_tests.py's run_dpu_test uses it)
    breakdown.csv   file, direction + the eight breakdown columns (last one named gpus): all files in one table
    blocksize.csv   file, block_size, compressed_bytes, space_saving, host_compress_s, gpu_run_s, lds_waves_per_cu
                    (counterpart of chart_compr_vs_blksize.py's input; lds_waves_per_cu = wavefronts per CU whose hash
                    table lives in LDS at that block size, snappy_hip_k1_lds_waves_per_cu)

Usage: python tools/run_sweeps.py --out DIR [--gpus 1,2,4,8] [--files a.txt,b.txt] [--mix-mib 256]
"""
import argparse
import csv
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "pim-compression_amd", "host", "dpu_snappy")
GOLDEN = os.path.join(ROOT, "tests", "golden")

BREAKDOWN_HEADER = ["prepare", "alloc", "load", "copy_in", "run", "copy_out", "free", "dpus"]   # run_tests.py:134
FIELDS = {"prepare": "Pre-processing time", "alloc": "Alloc time", "load": "Load time", "copy_in": "Copy in time",
          "run": "Host time", "copy_out": "Copy out time", "free": "Free time"}


def run_cli(args):
    r = subprocess.run([CLI] + args, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"dpu_snappy {' '.join(args)} failed: {r.stderr}")
    out = {}
    for key, label in FIELDS.items():
        m = re.search(re.escape(label) + r": ([0-9.]+)", r.stdout)
        out[key] = float(m.group(1)) if m else 0.0
    m = re.search(r"(?:Compressed|Decompressed) (\d+) bytes", r.stdout)
    out["bytes_out"] = int(m.group(1)) if m else 0
    m = re.search(r"Compression ratio: (-?[0-9.]+)", r.stdout)
    out["saving"] = float(m.group(1)) if m else 0.0
    out["gpu_kernel_s"] = sum(float(x) for x in re.findall(r"^GPU \d+: ([0-9.]+) s", r.stdout, re.M))
    return out


def make_mix(path, mib):
    sys.path.insert(0, os.path.join(ROOT, "pim-compression_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import silesia_mix
    tmp = path + ".xml"
    run_cli(["-i", os.path.join(GOLDEN, "xml.snappy"), "-o", tmp])        # CPU mode decodes the golden
    xml = np.fromfile(tmp, dtype=np.uint8)
    os.remove(tmp)
    unit = silesia_mix.build_unit(xml, seed=0)
    n = mib << 20
    reps = (n + unit.size - 1) // unit.size
    np.tile(unit, reps)[:n].tofile(path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--gpus", default="1")
    ap.add_argument("--files", default="")
    ap.add_argument("--mix-mib", type=int, default=256)
    ap.add_argument("--block-sizes", default="4096,8192,16384,32768,65535")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(CLI)])
    gpus = [int(g) for g in args.gpus.split(",")]
    work = tempfile.mkdtemp(prefix="snappy_sweep_")
    files = [f for f in args.files.split(",") if f] or [os.path.join(GOLDEN, n + ".txt") for n in
                                                        ("terror2", "plrabn12", "world192")]
    if args.mix_mib > 0:
        mix = os.path.join(work, f"silesia_mix_{args.mix_mib}MiB.bin")
        make_mix(mix, args.mix_mib)
        files.append(mix)

    ref_speedup = {"compress": [["version", "time", "dpus"], ["host", "1", "0"]],
                   "decompress": [["version", "time", "dpus"], ["host", "1", "0"]]}
    with open(os.path.join(args.out, "speedup.csv"), "w", newline="") as fs, \
            open(os.path.join(args.out, "breakdown.csv"), "w", newline="") as fb:
        ws, wb = csv.writer(fs), csv.writer(fb)
        ws.writerow(["file", "bytes", "direction", "gpus", "host_s", "gpu_kernel_s", "gpu_total_s", "speedup_kernel",
                     "speedup_total"])
        wb.writerow(["file", "direction", "prepare", "alloc", "load", "copy_in", "run", "copy_out", "free", "gpus"])
        for f in files:
            name, size = os.path.basename(f), os.path.getsize(f)
            comp = os.path.join(work, name + ".snappy")
            host_c = run_cli(["-c", "-i", f, "-o", comp])
            host_d = run_cli(["-i", comp, "-o", os.path.join(work, name + ".host_out")])
            per_file = {"compress": [BREAKDOWN_HEADER], "decompress": [BREAKDOWN_HEADER]}
            for g in gpus:
                gc = run_cli(["-d", "-g", str(g), "-c", "-i", f, "-o", comp + ".gpu"])
                gd = run_cli(["-d", "-g", str(g), "-i", comp, "-o", os.path.join(work, name + ".gpu_out")])
                if open(comp, "rb").read() != open(comp + ".gpu", "rb").read() or \
                        open(f, "rb").read() != open(os.path.join(work, name + ".gpu_out"), "rb").read():
                    raise RuntimeError(f"parity failure on {name} with {g} GPU(s)")
                for direction, host, gpu in (("compress", host_c, gc), ("decompress", host_d, gd)):
                    total = gpu["copy_in"] + gpu["run"] + gpu["copy_out"]
                    ws.writerow([name, size, direction, g, f"{host['run']:.6f}", f"{gpu['gpu_kernel_s']:.6f}", f"{total:.6f}",
                                 f"{host['run'] / max(gpu['gpu_kernel_s'], 1e-9):.2f}", f"{host['run'] / max(total, 1e-9):.2f}"])
                    wb.writerow([name, direction] + [f"{gpu[k]:.6f}" for k in FIELDS] + [g])
                    per_file[direction].append([f"{gpu[k]:.6f}" for k in FIELDS] + [g])
                    overhead = sum(gpu[k] for k in FIELDS if k != "run")          # run_tests.py:97: host / (dpu + sum(overhead))
                    ref_speedup[direction].append([name, f"{host['run'] / max(gpu['run'] + overhead, 1e-9):.4f}", g])
            stem = os.path.splitext(name)[0]
            for direction, rows in per_file.items():
                with open(os.path.join(args.out, f"{stem}_{direction}ion_breakdown.csv"), "w", newline="") as fp:
                    csv.writer(fp).writerows(rows)
    for direction, rows in ref_speedup.items():
        with open(os.path.join(args.out, f"{direction}ion_speedup_dpu.csv"), "w", newline="") as fp:
            csv.writer(fp).writerows(rows)

    with open(os.path.join(args.out, "blocksize.csv"), "w", newline="") as fz:
        wz = csv.writer(fz)
        wz.writerow(["file", "block_size", "compressed_bytes", "space_saving", "host_compress_s", "gpu_run_s", "lds_waves_per_cu"])
        sys.path.insert(0, os.path.join(ROOT, "pim-compression_amd"))
        import snappy_hip_binding as shb
        for f in files:
            name = os.path.basename(f)
            for bs in [int(b) for b in args.block_sizes.split(",")]:
                outp = os.path.join(work, f"{name}.b{bs}")
                h = run_cli(["-c", "-b", str(bs), "-i", f, "-o", outp])
                g = run_cli(["-d", "-c", "-b", str(bs), "-i", f, "-o", outp + ".gpu"])
                if open(outp, "rb").read() != open(outp + ".gpu", "rb").read():
                    raise RuntimeError(f"parity failure on {name} at block size {bs}")
                wz.writerow([name, bs, h["bytes_out"], f"{h['saving']:.6f}", f"{h['run']:.6f}", f"{g['gpu_kernel_s']:.6f}",
                             shb.k1_lds_waves_per_cu(bs)])
    print("wrote", args.out)


if __name__ == "__main__":
    main()
