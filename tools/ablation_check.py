#!/usr/bin/env python3
"""GPU check of the ablation build (libsnappy_hip_ablation.so): every non-default K1 / K2 form is bit-exact with the oracle.
Run by tests/test_gpu_ablation.py in a process of its own (the product library must not be loaded beside it).
Usage: python tools/ablation_check.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "pim-compression_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np
import torch

import build_ablation
import datagen
import oracle_lib as oracle
import snappy_hip_binding as shb

F0 = {"SNAPPY_HIP_K1_FORM": "0", "SNAPPY_HIP_K1_FILTER": "0", "SNAPPY_HIP_K1_FORM_LDS": "0"}
TINY_HYBRID = {"SNAPPY_HIP_LDS_WAVES": "5", "SNAPPY_HIP_GT_WAVES": "11", "SNAPPY_HIP_HYBRID_MIN_BLOCKS": "1"}
VARIANTS = [
    {"SNAPPY_HIP_COMPRESS_VARIANT": "4"}, {"SNAPPY_HIP_COMPRESS_VARIANT": "4", "SNAPPY_HIP_LANES_PER_BLOCK": "16"},
    {"SNAPPY_HIP_COMPRESS_VARIANT": "5"}, {"SNAPPY_HIP_COMPRESS_VARIANT": "5", "SNAPPY_HIP_GROUP_WAVES": "3"},
    # windowed form: serial probes and look-ahead widths
    {**F0, "SNAPPY_HIP_K1_AHEAD": "0", "SNAPPY_HIP_K1_AHEAD_LDS": "0"},
    {**F0, "SNAPPY_HIP_K1_AHEAD": "8"}, {**F0, "SNAPPY_HIP_K1_AHEAD": "16"}, {**F0},
    {**F0, "SNAPPY_HIP_LDS_WAVES": "1024"},
    {**F0, "SNAPPY_HIP_K1_AHEAD_LDS": "8", "SNAPPY_HIP_COMPRESS_VARIANT": "1"},
    {**F0, **TINY_HYBRID, "SNAPPY_HIP_K1_AHEAD": "16"},
    {**F0, "SNAPPY_HIP_K1_FILTER": "1"},
    # masked form
    {"SNAPPY_HIP_K1_FORM": "1", "SNAPPY_HIP_K1_FILTER": "0", "SNAPPY_HIP_K1_AHEAD": "32"},
    {"SNAPPY_HIP_K1_FORM": "1", "SNAPPY_HIP_K1_FORM_LDS": "1", **TINY_HYBRID},
    {"SNAPPY_HIP_K1_FORM": "1"},
    {"SNAPPY_HIP_K1_FORM_LDS": "1", "SNAPPY_HIP_K1_AHEAD_LDS": "32", "SNAPPY_HIP_COMPRESS_VARIANT": "1"},
    # bulk form: without filter, narrower chunks, class filter
    {"SNAPPY_HIP_K1_FILTER": "0"}, {"SNAPPY_HIP_K1_FILTER": "0", "SNAPPY_HIP_K1_AHEAD": "32"},
    {"SNAPPY_HIP_K1_AHEAD": "32", **TINY_HYBRID}, {"SNAPPY_HIP_K1_FILTER": "2"}, {"SNAPPY_HIP_K1_FILTER": "2", **TINY_HYBRID},
    # K2: output window in LDS, both forms concurrently
    {"SNAPPY_HIP_DECOMPRESS_VARIANT": "0"}, {"SNAPPY_HIP_DECOMPRESS_VARIANT": "2", "SNAPPY_HIP_HYBRID_MIN_BLOCKS": "1"},
    {"SNAPPY_HIP_K2_BATCH": "0"},
    # round 2's two-wavefront LDS-table workgroups (csrc/ablation/k1_pair_kernel.hpp): alone, beside global-table wavefronts
    {"SNAPPY_HIP_PAIR_PER_CU": "4", "SNAPPY_HIP_GT_WAVES": "0"}, {"SNAPPY_HIP_PAIR_PER_CU": "3"},
    {"SNAPPY_HIP_PAIR_PER_CU": "1", "SNAPPY_HIP_GT_WAVES": "64"},
    # round 3's duo form (csrc/ablation/k1_duo_form.hpp: parser + mate wavefront per LDS-table block)
    {"SNAPPY_HIP_K1_STREAM": "5"}, {"SNAPPY_HIP_K1_STREAM": "5", "SNAPPY_HIP_COMPRESS_VARIANT": "1"},
    {**TINY_HYBRID, "SNAPPY_HIP_K1_STREAM": "5"},
]


def to_dev(data):
    t = torch.zeros(len(data) + 16, dtype=torch.uint8, device="cuda")
    if len(data):
        t[:len(data)] = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    return t


def main():
    shb.LIB_PATH = build_ablation.build()
    text = open(os.path.join(ROOT, "tests", "golden", "plrabn12.txt"), "rb").read()
    cases = [open(os.path.join(ROOT, "tests", "golden", "world192.txt"), "rb").read(),
             datagen.text_random_interleave(text, 300_000), datagen.records(200_000), datagen.zeros(70_000),
             datagen.lz_structured(200_000, 9), datagen.random_bytes(100_000)]
    bad = 0
    for env in VARIANTS:
        os.environ.update(env)
        for data in cases:
            for bs in (32768, 4097, 65535):
                ref = oracle.compress(data, bs)
                got = bytes(shb.compress_resident(to_dev(data), bs, n=len(data)).cpu().numpy())
                st, out = shb.decompress_resident(to_dev(ref), stream_len=len(ref))
                if got != ref or st != 0 or bytes(out.cpu().numpy()) != data:
                    bad += 1
                    print("MISMATCH", env, len(data), bs, flush=True)
        for k in env:
            os.environ.pop(k, None)
    print(f"ablation variants {len(VARIANTS)} bad {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
