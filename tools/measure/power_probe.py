"""Board power and shader clock (rocm-smi) while one kernel family runs back to
back: the correlation kernel vs a memory-bound mesh step vs idle."""
import sys, time, os, subprocess, threading, re, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sofima_amd import flow_field, mesh
from bench import synth_pair


def sample(stop, out):
  while not stop.is_set():
    try:
      r = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--json'], capture_output=True,
                         text=True, timeout=5).stdout
      d = json.loads(r)
      card = next(iter(d.values()))
      pw = [float(v) for k, v in card.items() if 'ower' in k and re.match(r'^[0-9.]+$', str(v))]
      sclk = [v for k, v in card.items() if 'sclk' in k.lower()]
      out.append((time.perf_counter(), pw, sclk))
    except Exception as e:  # keep sampling
      out.append((time.perf_counter(), repr(e)[:80], None))
    time.sleep(0.1)


def run(name, fn, seconds=4.0):
  fn(); torch.cuda.synchronize()
  stop, out = threading.Event(), []
  th = threading.Thread(target=sample, args=(stop, out)); th.start()
  t0 = time.perf_counter(); n = 0
  while time.perf_counter() - t0 < seconds:
    fn(); n += 1
    if n % 4 == 0: torch.cuda.synchronize()
  torch.cuda.synchronize(); stop.set(); th.join()
  pws = [p[0] for _, p, _ in out if isinstance(p, list) and p]
  clk = [c for _, _, c in out if c]
  print('%-34s %4d calls, %3d samples, power W: median %s max %s; sclk samples: %s' % (
      name, n, len(out), np.median(pws) if pws else None, max(pws) if pws else None, clk[len(clk) // 2] if clk else None),
      flush=True)
  if not pws and out: print('   raw sample:', out[0])


pre, post = synth_pair(8192, 1002)
a = torch.from_numpy(pre).cuda(); b = torch.from_numpy(post).cuda()
calc = flow_field.JAXMaskedXCorrWithStatsCalculator()
run('idle (sleep)', lambda: time.sleep(0.05), 2.0)
run('flow 8192^2 pairs (int8 MFMA)', lambda: calc.flow_field(a, b, 160, 40, batch_size=1024, device_output=True))
shape = (2, 1, 2048, 2048)
rng = np.random.default_rng(0)
pv = torch.from_numpy((rng.standard_normal(shape) * 5).astype(np.float32)).cuda()
x = torch.zeros(shape, device='cuda')
cfg = mesh.IntegrationConfig(dt=0.001, gamma=0.0, k0=0.01, k=0.1, stride=(40, 40), num_iters=200, max_iters=200,
                             stop_v_max=1e-9, dt_max=1000, start_cap=0.01, final_cap=10, prefer_orig_order=True)
run('mesh 2048^2 steps (HBM bound)', lambda: mesh.relax_mesh(x, pv, cfg))
big = torch.empty(1 << 30, dtype=torch.uint8, device='cuda')
run('fill 1 GiB (pure HBM writes)', lambda: big.zero_())
