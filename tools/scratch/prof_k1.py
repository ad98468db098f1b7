import os, sys, ctypes
ROOT = "/root/repo"
for p in (os.path.join(ROOT, "pim-compression_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import silesia_mix
import snappy_hip_binding as shb
shb.LIB_PATH = os.path.join(ROOT, "pim-compression_amd", "libsnappy_hip_prof.so")
L = shb.lib()
L.snappy_hip_debug_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
mib = int(sys.argv[1]); n = mib << 20
xs = np.frombuffer(open(os.path.join(ROOT, "tests/golden/xml.snappy"), "rb").read(), dtype=np.uint8).copy()
st, d_xml = shb.decompress_resident(torch.from_numpy(xs).cuda())
unit = silesia_mix.build_unit(d_xml.cpu().numpy(), seed=0)
d_in = silesia_mix.container_from_unit(torch.from_numpy(unit).cuda(), n)
ws = shb.CompressWorkspace(n, 32768)
names = ["t_total", "blocks", "t_gather", "n_gather", "t_serial", "n_serial", "n_probe", "t_match", "n_match", "t_lit", "t_win", "n_win", "t_ext", "n_ext"]
for cfg in sys.argv[2:]:
    kv = dict(x.split("=") for x in cfg.split(","))
    for k, v in kv.items(): os.environ[k] = v
    shb.compress_blocks(d_in, n, ws); torch.cuda.synchronize()
    L.snappy_hip_debug_prof(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); shb.compress_blocks(d_in, n, ws); e1.record(); torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    L.snappy_hip_debug_prof(out, 0)
    p = dict(zip(names, list(out)))
    tot = p["t_total"]
    print(f"== {cfg}: {e0.elapsed_time(e1):.2f} ms; blocks {p['blocks']}; cycles/block {tot/max(p['blocks'],1):.0f}; probes/block {p['n_probe']/max(p['blocks'],1):.0f}")
    def pct(x): return 100.0 * x / max(tot, 1)
    print(f"   gather  {pct(p['t_gather']):5.1f}%  n={p['n_gather']}  {p['t_gather']/max(p['n_gather'],1):.0f} cyc each")
    print(f"   serialf {pct(p['t_serial']):5.1f}%  n={p['n_serial']}  {p['t_serial']/max(p['n_serial'],1):.0f} cyc each")
    print(f"   match   {pct(p['t_match']):5.1f}%  n={p['n_match']}  {p['t_match']/max(p['n_match'],1):.0f} cyc each (incl. extend {pct(p['t_ext']):.1f}% n={p['n_ext']} {p['t_ext']/max(p['n_ext'],1):.0f} each)")
    print(f"   literal {pct(p['t_lit']):5.1f}%")
    print(f"   window  {pct(p['t_win']):5.1f}%  n={p['n_win']}  {p['t_win']/max(p['n_win'],1):.0f} cyc each")
    rest = tot - p['t_gather'] - p['t_serial'] - p['t_match'] - p['t_lit'] - p['t_win']
    print(f"   rest    {pct(rest):5.1f}%  = {rest/max(p['n_probe'],1):.0f} cyc per probe", flush=True)
    for k in kv: os.environ.pop(k, None)
