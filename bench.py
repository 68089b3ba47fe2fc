#!/usr/bin/env python3
"""bench.py -- slides/sec of the MADELEINE cross-stain pretrain step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W           (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" = zero_grad + encoder fwd + losses + bwd + AdamW step, train mode (dropout on), on DEVICE-RESIDENT
synthetic bags: per rank B=32 slides x M stains x N=4096 patches x D=512 fp32
(weak scaling: per-GPU work fixed).  N=1 workload = BASELINE.json configs[1] ("c2": 2 stains, ABMIL pool +
global InfoNCE), the configuration the metric is quoted on.  Rank 0 prints ONE JSON line.

`python bench.py --gpus N` with no launcher in the environment re-executes itself as N ranks under torch.distributed.run (loopback
rendezvous); under a launcher (RANK / WORLD_SIZE set) it joins the group it was given.

Output: rank 0 prints a `BENCH_DETAIL {...}` line (the full record: secondary legs, variants, provenance; also written to
gpurun_out/bench_detail.json) and then, as the LAST stdout line, ONE compact JSON line (<= 4 KB -- `compact_line`) with the contract keys plus
  roofline        -- A3 softmax-pool forward kernel (the HBM-bound kernel north_star targets at >= 60 %): algorithmic bytes
                     (8,208 B/token + 8 KiB/bag) / the dispatches' own start-stop HIP events inside the timed region; `traffic` from
                     two in-run rocprofv3 PMC passes.
  cpu_baseline    -- the CPU oracle (oracle/restatement.py, kind "port") timed on this box's host cores on the full 32-slide step
                     (rank 0, N=1 only).
  kernels         -- every kernel family of the headline step: [ms per call, fraction of peak] -- the dense MFMA peak of the instruction
                     it runs on (split mode: 3 fp16 MFMA FLOPs per algorithmic fp32 FLOP against 2.5 PFLOP/s) or the 8 TB/s HBM peak.
  roofline_mfma   -- A2 gate kernels (fwd + dX + dW), the time-dominant contractions.
  secondary_ms_per_step -- step times of the secondary legs (bf16 mode, config 3, one rank of configs 4 / 5, PCIe-inclusive feed).
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace


import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MODS5 = ["HE", "HER2", "PGR", "KI67", "ER"]
# torch.optim.AdamW(params, lr) as the reference builds it (setup_components.py:196: betas, eps, weight_decay 0.01 = torch defaults), in
# torch's single-kernel `fused` implementation: same update, one launch instead of ~8 multi-tensor launches (23.7 -> 23.1 ms per
# config-2 step in a same-box A/B).  BENCH_FOREACH_ADAMW=1 selects torch's default (foreach) implementation.
FUSED_ADAMW = not os.environ.get("BENCH_FOREACH_ADAMW")
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F32_MFMA_PEAK_TF = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak

CONFIGS = {
    # name: (B per rank, M, N, D, use_got, stain_encoding)
    "c1": (4, 2, 256, 512, False, False),
    "c2": (32, 2, 4096, 512, False, False),
    "c3": (32, 5, 4096, 512, True, False),
    # config 5 (ragged stress): N_i ~ U{1024..16384} per bag, d=768, stain-encoding tokens, full global+local loss
    "c5": (32, 5, 0, 768, True, True),
}


def make_cfg(M, D):
    return SimpleNamespace(MODALITIES=MODS5[:M], wsi_encoder="abmil", patch_embedding_dim=D,
                           wsi_encoder_hidden_dim=512, activation="softmax", n_heads=4)


def usable_cores() -> int:
    """Host cores this process may actually use: min(affinity mask, cgroup v2 cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def pool_traffic_from_profiles(tokens_c2=64 * 4096):
    """HBM bytes per launch of the A3 forward from the committed PMC passes (profiles/*_pool_pmc_{fetch,write}.txt:
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`, separate passes, same geometry as config 2).  FETCH_SIZE is in
    KiB and on gfx950 counts half of a wide coalesced read stream (MI355X_MICROARCH.md, HBM section) -> x2."""
    import glob
    import re

    def avg(kind, counter):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_pool_pmc_{kind}.txt")))
        if not files:
            return None
        tot = 0.0
        for kern in ("pool_partial_kernel", "pool_combine_kernel"):
            m = re.search(kern + r"\S*\s+" + counter + r"\s+\d+\s+[0-9.]+\s+([0-9.]+)", open(files[-1]).read())
            if not m:
                return None
            tot += float(m.group(1))
        return tot
    f, w = avg("fetch", "FETCH_SIZE"), avg("write", "WRITE_SIZE")
    if f is None or w is None:
        return None
    return int((2.0 * f + w) * 1024)


def measure_pool_traffic(timeout_s=150, image=False):
    """HBM bytes per launch of the A3 forward MEASURED IN THIS RUN: two short rocprofv3 passes (`--pmc FETCH_SIZE`, `--pmc
    WRITE_SIZE`; separate passes, kernel-trace only -- MI355X_MICROARCH.md, HBM section) over tools/prof_kernels.py --only pool
    (config-2 geometry), read back from the rocpd database.  FETCH_SIZE is in KiB and on gfx950 counts half of a wide coalesced
    read stream -> x2.  Returns (bytes, provenance) or (None, reason)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="mdl_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run([exe, "--pmc", counter, "--kernel-trace", "-d", d, "--", sys.executable,
                                os.path.join(ROOT, "tools", "prof_kernels.py"), "--iters", "2", "--only", "pool"] + (["--image"] if image else []),
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            dbs = glob.glob(os.path.join(d, "*", "*.db"))
            if r.returncode != 0 or not dbs:
                return None, "rocprofv3 --pmc %s failed (rc %d)" % (counter, r.returncode)
            c = sqlite3.connect(dbs[0])
            rows = c.execute("""select s.kernel_name, avg(e.value) from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
                                join rocpd_kernel_dispatch d on d.event_id = e.event_id
                                join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                                where p.name = ? group by s.kernel_name""", (counter,)).fetchall()
            tot = sum(v for k, v in rows if ("pool_partial_kernel" in k or "pool_combine_kernel" in k))
            if tot <= 0:
                return None, "no %s samples for the pool kernels" % counter
            vals[counter] = tot
        except Exception as e:  # timeout, sqlite schema, ...
            return None, "%s: %s" % (type(e).__name__, e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024), \
        "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (2 passes) over tools/prof_kernels.py --only pool%s; 2 x FETCH + WRITE" % (
            " --image" if image else "")


class PowerSampler:
    """Socket power (W) and shader clock (MHz) from the amdgpu hwmon files (power1_input / power1_average, freq1_input), sampled by a
    background thread -- cheap file reads, no rocm-smi process.  Used over a step loop OUTSIDE the timed region."""

    def __init__(self, period_s=0.05):
        import glob
        self.period = period_s
        self.power = self.freq = self.cap = None
        # the hwmon directory of the device HIP runs on (a host can hold many amdgpu cards): match its PCI address
        want = None
        try:
            pr = torch.cuda.get_device_properties(torch.cuda.current_device())
            want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        except Exception:  # noqa: BLE001
            pass
        cands = []
        for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            for name in ("power1_input", "power1_average"):
                if os.path.exists(os.path.join(d, name)) and os.path.exists(os.path.join(d, "freq1_input")):
                    addr = os.path.basename(os.path.realpath(os.path.join(d, "..", "..")))
                    cands.append((addr, d, name))
                    break
        pick = [c for c in cands if want and c[0].startswith(want)]
        self.matched_by = "pci address %s" % want if pick else ("highest power reading" if cands else None)
        if not pick and cands:   # fall back to the card that draws the most power right now (the caller is running a step loop)
            def rd(c):
                try:
                    return self._read(os.path.join(c[1], c[2]))
                except (OSError, ValueError):
                    return -1.0
            pick = [max(cands, key=rd)]
        if pick:
            _, d, name = pick[0]
            self.power, self.freq = os.path.join(d, name), os.path.join(d, "freq1_input")
            cap = os.path.join(d, "power1_cap")
            self.cap = cap if os.path.exists(cap) else None
        self.samples = []
        self._stop = None

    @staticmethod
    def _read(path):
        with open(path) as f:
            return float(f.read().strip())

    def __enter__(self):
        import threading
        if self.power is None:
            return self
        self._stop = threading.Event()

        def run():
            while not self._stop.is_set():
                try:
                    self.samples.append((time.perf_counter(), self._read(self.power) / 1e6, self._read(self.freq) / 1e6))
                except (OSError, ValueError):
                    pass
                self._stop.wait(self.period)
        self._th = threading.Thread(target=run, daemon=True)
        self._th.start()
        return self

    def __exit__(self, *exc):
        if self._stop is not None:
            self._stop.set()
            self._th.join()
        return False

    def summary(self, skip_s=0.3):
        if not self.samples:
            return {"source": "amdgpu hwmon not readable on this box"}
        t0 = self.samples[0][0]
        xs = [x for x in self.samples if x[0] - t0 >= skip_s] or self.samples
        pw, fq = sorted(x[1] for x in xs), sorted(x[2] for x in xs)
        med = lambda v: v[len(v) // 2]   # noqa: E731
        out = {"source": "amdgpu hwmon (%s, freq1_input; card matched by %s), %d samples at %.0f ms over the step loop" % (
                   os.path.basename(self.power), self.matched_by, len(xs), 1e3 * self.period),
               "socket_power_W": {"median": round(med(pw), 1), "min": round(pw[0], 1), "max": round(pw[-1], 1)},
               "sclk_MHz": {"median": round(med(fq)), "min": round(fq[0]), "max": round(fq[-1])}}
        if self.cap:
            try:
                out["power_cap_W"] = round(self._read(self.cap) / 1e6, 1)
            except (OSError, ValueError):
                pass
        return out


def cpu_baseline(B_sample, M, N, D, use_got, steps=2, train=True):
    """Times the CPU oracle's full step (fwd + losses + bwd + AdamW; train: dropout on, as the reference trains) on B_sample slides."""
    from oracle import restatement as R
    threads = usable_cores()
    torch.set_num_threads(threads)
    mods = MODS5[:M]
    sd = R.make_params(M, D, 4, False, seed=42)
    params = [v.requires_grad_() for v in sd.values()]
    opt = torch.optim.AdamW(params, lr=1e-4)
    g = torch.Generator().manual_seed(1234)
    feats = torch.randn(B_sample, M, N, D, generator=g)
    labels = torch.ones(B_sample, M)
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        opt.zero_grad()
        pre, gate = R.random_keep_masks(B_sample * M, N, 4, g) if train else (None, None)
        loss, flag, _ = R.pretrain_step_loss(feats, labels, sd, mods, 0.001, True, use_got=use_got, pre_keep=pre,
                                             gate_keep=gate)
        loss.backward()
        opt.step()
        dt = time.perf_counter() - t0
        if it > 0:
            times.append(dt)
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(B_sample / med, 4), "unit": "slides/s", "cores": threads, "kind": "port",
            "sample": f"{B_sample} of the step's slides x {M} stains x {N} x {D}, full step (fwd+loss+bwd+AdamW), "
                      f"{'train-mode dropout' if train else 'dropout off'}, median of {steps} after 1 warm-up; {med:.2f} s/step"}


def measure_leg(MF, stepf, steps, warmup=2, prof_steps=3):
    """A secondary leg: `steps` timed steps with NO per-launch events (cuda synchronize on both sides), then a short profiled pass for
    the per-kernel figures.  -> (seconds for the timed steps, last loss, {kernel: (avg ms, calls)}, declared work)."""
    for _ in range(warmup):
        stepf()
    torch.cuda.synchronize()
    MF.TIMER = None
    dev = torch.cuda.current_device()
    stable = []
    for _attempt in range(8):
        # steady state only: a leg that follows torch.cuda.empty_cache() can still be growing its pools, and a hipMalloc right after tens
        # of GiB were freed waits for the driver to scrub them (seen: 2-6x the step time in one of four processes; round 6: the ragged
        # bf16 leg still allocating through three attempts -> 163 ms instead of 61-67) -- time again then; what is reported says so
        a0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = stepf()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        measure_leg.last_allocs = torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - a0
        if measure_leg.last_allocs == 0:
            # a second allocation-free pass, the faster of the two reported: single steps of the ragged legs come out 20-35 ms long now
            # and then (66 / 73 / 88 ms for the same c5 bf16 step in back-to-back passes, tools/debug/dbg_c5_bf16_leg.py) -- host-side
            # (pinned staging blocks still in flight -> a fresh hipHostMalloc), not a property of the kernels
            stable.append(el)
            if len(stable) == 2:
                break
    if stable:
        el = min(stable)
    MF.TIMER = MF.KernelTimer()
    try:
        for _ in range(prof_steps):
            stepf()
        prof, work = MF.TIMER.report(), MF.TIMER.work()
    finally:
        MF.TIMER = None
    return el, loss, prof, work, prof_steps


measure_leg.last_allocs = 0


def secondary_c3_leg(dev, D, MF, InfoNCE, MADELEINE, steps=5, warmup=2, all_present=False, skip_absent=False):
    """BASELINE configs[2] as a short secondary measurement beside the headline: 32 slides x 5 stains (ACROBAT presence rates,
    absent stain = all-zero bag) x 4096 x 512, global InfoNCE + local GOT (IPOT Wasserstein + Gromov-Wasserstein, n = k <= 32
    tokens), AdamW, train mode.  The headline workload (c2) has no GOT: this leg is where the GOT kernels are timed.
    all_present: SURVEY 8(d)'s second mask variant -- every case carries every stain (k = 32 for all four GOT problems, no zero bags).
    skip_absent: SURVEY 8(f) N4 -- the all-zero bag of an absent stain (wsi_dataset.py:66; no loss term reads its outputs, trainer.py:27-29)
    is encoded once instead of once per case (MADELEINE.skip_absent_stains; never the default: the reference encodes them all)."""
    B, M, N, Dm, _, _ = CONFIGS["c3"]
    mods = MODS5[:M]
    torch.manual_seed(42)
    model = MADELEINE(make_cfg(M, Dm)).to(dev).train()
    model.skip_absent_stains = bool(skip_absent)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=FUSED_ADAMW)
    gen = torch.Generator(device=dev).manual_seed(1234)
    feats = torch.randn(B, M, N, Dm, device=dev, generator=gen)
    rates = torch.tensor([1.0, 0.46, 0.73, 0.73, 0.73])
    labels = (torch.rand(B, M, generator=torch.Generator().manual_seed(77)) < rates).float()
    if all_present:
        labels = torch.ones(B, M)
    labels[:, 0] = 1
    feats = feats * labels.to(dev)[:, :, None, None]
    data = {"feats": feats, "modality_labels": labels}
    crit = InfoNCE(temperature=0.001)
    largs = SimpleNamespace(global_loss="info-nce", symmetric_cl=True, local_loss_weight=1.0)

    def step():
        opt.zero_grad(set_to_none=True)
        embs, toks = model(data, device=dev)
        loss, _ = D.calculate_losses_dp(mods[1:], crit, MF.HipGotImpl, embs, toks, labels[:, 1:], largs)
        loss.backward()
        opt.step()
        return loss

    el, loss, prof, work, psteps = measure_leg(MF, step, steps, warmup)
    k = [int(labels[:, s].sum()) for s in range(1, M)]
    return {"value": round(B * steps / el, 3), "unit": "slides/s", "ms_per_step": round(1e3 * el / steps, 3), "steps": steps,
            "workload": f"c3{' (all stains present)' if all_present else ''}: {B} slides x {M} stains (cases per stain {k}) x {N} x {Dm}, "
                        f"InfoNCE + GOT, train mode, AdamW" + (f"; absent-stain zero bags encoded once ({int(labels.sum())} of {B * M} bags "
                                                               f"present)" if skip_absent else ""),
            "final_loss": float(loss.detach()),
            "kernel_ms": {n: round(v[0], 4) for n, v in prof.items()}, "kernel_calls_per_step": {n: v[1] // psteps for n, v in prof.items()},
            "kernel_roofline": kernel_rooflines(prof, work, "float32", MF.gemm_mode())}


F16_MFMA_PEAK_TF = 2500.0   # MI355X_MICROARCH.md: dense fp16 / bf16 matrix peak (v_mfma_f32_32x32x16_f16 / _bf16)
# What the matrix cores SUSTAIN under the 1400 W package cap with register-only operands of random content (no LDS / HBM traffic at all;
# tools/micro/mfma_power.hip, profiles/r04h_mfma_power_by_operand_content.txt): the nominal peak needs operands with few toggling bits
# (2.2-2.4 PF measured with small-integer operands); fp32 MFMA is not power-limited (151.5 of 157.3).
MFMA_SUSTAINED_RANDOM_TF = {"f16": 1660.0, "bf16": 1820.0, "f32": 151.5}


def kernel_rooflines(prof, work, precision="float32", gemm_mode="fp32"):
    """Achieved rate of every kernel family that declared its algorithmic work (functional.KernelTimer.work): TFLOP/s against the
    dense MFMA peak of the dtype for the contractions, GB/s against the 8 TB/s HBM3E peak for the bandwidth-bound passes.  Event
    times are un-profiled clocks (rocprofv3 lowers the clocks: the durations in profiles/ are 7-12 % longer)."""
    # split GEMM mode (fp32 precision): every algorithmic fp32 FLOP costs three fp16 MFMA FLOPs (ah bh + ah bl + al bh): `achieved` stays
    # algorithmic (fp32-equivalent), `frac` = 3 x achieved / the fp16 matrix peak = the matrix-core utilisation of the kernels
    split = precision == "float32" and gemm_mode == "split"
    mpeak = (F16_MFMA_PEAK_TF if split else F32_MFMA_PEAK_TF) if precision == "float32" else 2500.0
    mult = 3.0 if split else 1.0
    out = {}
    for name, (kind, per_call) in sorted(work.items()):
        if name not in prof or prof[name][0] <= 0:
            continue
        ms = prof[name][0]
        if kind == "flop":
            ach = per_call / (ms * 1e-3) / 1e12
            out[name] = {"bound": "mfma", "avg_ms": round(ms, 4), "achieved": round(ach, 1), "unit": "TFLOP/s", "peak": mpeak,
                         "frac": round(mult * ach / mpeak, 4)}
            if split:
                out[name]["raw_mfma_tflops"] = round(mult * ach, 1)
        else:
            ach = per_call / (ms * 1e-3) / 1e9
            out[name] = {"bound": "hbm", "avg_ms": round(ms, 4), "achieved": round(ach, 1), "unit": "GB/s", "peak": HBM_PEAK_GBS,
                         "frac": round(ach / HBM_PEAK_GBS, 4)}
    return out


def committed_mfma_busy():
    """Matrix-core occupancy of the split-engine kernel families from the COMMITTED counter passes (profiles/*_split_mfma_busy.json,
    written by tools/collect_profiles.sh <tag> split + tools/pmc_mfma_busy.py: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x
    GRBM_GUI_ACTIVE), separate --pmc passes over tools/prof_split.py at config-2 geometry).  Counters cannot be collected inside a timed
    run; the figure sits beside `frac` with its provenance.  Returns {family: {...}} (empty when no file is committed)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_split_mfma_busy.json")))
    if not files:
        return {}
    try:
        d = json.load(open(files[-1]))
    except (OSError, ValueError):
        return {}
    src = os.path.relpath(files[-1], ROOT)

    def pick(*needles):
        ks = [v for k, v in d.items() if all(n in k for n in needles) and v.get("mfma_busy")]
        if not ks:
            return None
        w = [(v["avg_us_under_pmc"] or 0) * (v["dispatches"] or 0) for v in ks]
        tot = sum(w) or 1.0
        return {"mfma_busy": round(sum(v["mfma_busy"] * wi for v, wi in zip(ks, w)) / tot, 4),
                "effective_clock_GHz_under_pmc": round(sum((v.get("effective_clock_GHz") or 0) * wi for v, wi in zip(ks, w)) / tot, 3),
                "source": src}
    fams = {"gate_fwd": pick("gate_fwd_kernel"), "gate_bwd_gemm": pick("sp_gate_d"), "linear_fwd": pick("sp_nt_kernel"),
            "linear_bwd": pick("sp_", "n_kernel")}
    return {k: v for k, v in fams.items() if v}


def emulated_rank_loss(D, MF, crit, largs, mods, embs, toks, labels, lab_g, noise, world):
    """The loss ONE rank of a `world`-rank job backpropagates (distributed.calculate_losses_dp's formula), with the other ranks' share of the
    all-gathered payload EMULATED instead of gathered -- no collective runs:
      * slide embeddings of the global batch = [local | detached copies of the local ones + noise[m]] ([world * B_l, 1, 512] per modality);
      * presence labels of the global batch = lab_g ([world * B_l, M], rows 0..B_l-1 = this rank's `labels`);
      * GOT: this rank's cases only, at the GLOBAL token count n = min(k_global, 256) (loss.py:282: randperm(k) with k = the participating
        cases of the global batch), thresholds = this rank's own extrema standing in for the gathered ones, weighted by `world`
        (the gradient MEAN over ranks then equals the global-batch gradient).
    loss = replicated global InfoNCE + world x local GOT sum.  tests/test_bench_path_gpu.py checks it against the oracle on the
    concatenated global batch (8 emulated ranks x 2 cases)."""
    from madeleine_amd.trainer import calculate_losses
    M = len(mods)
    dev = toks["HE"].device
    k_g = [int(lab_g[:, s].sum()) for s in range(1, M)]
    problems, kept = [], []
    for s_idx, stain in enumerate(mods[1:]):
        n = min(k_g[s_idx], 256)
        rows = labels[:, 1 + s_idx].bool().nonzero(as_tuple=True)[0]
        if k_g[s_idx] <= 1 or rows.numel() == 0:
            continue
        rows = MF.h2d(rows, dev)
        kept.append(s_idx)
        problems.append((toks["HE"][:, :n, :, s_idx].index_select(0, rows).float().contiguous(),
                         toks[stain][:, :n].index_select(0, rows).float().contiguous()))
    embs_g = {}
    for m in mods:
        loc = embs[m][..., 0] if m == "HE" else embs[m]                      # [B,1,512]
        oth = loc.detach().repeat(world - 1, 1, 1) + noise[m]
        full = torch.cat([loc, oth])
        embs_g[m] = full.unsqueeze(3).expand(-1, -1, -1, M - 1) if m == "HE" else full
    outs = None
    if problems:
        ext = D.got_local_extrema(problems, MF.HipGotImpl)      # "gathered" extrema: this rank's own stand in for the global ones
        outs = D.got_multi(problems, MF.HipGotImpl, None, extrema=ext)     # queued before the InfoNCE section, as calculate_losses_dp does
    loss_g, flag = calculate_losses(mods[1:], crit, None, None, embs_g, None, lab_g[:, 1:], largs)
    if outs is None:
        return loss_g
    return loss_g + float(world) * largs.local_loss_weight * (outs[:, 0] + outs[:, 1]).sum()


def secondary_rank_leg(dev, D, MF, InfoNCE, MADELEINE, world=8, steps=4, warmup=2, config="c3", all_present=False, bf16=False):
    """What ONE rank of an 8 x MI355X configuration (BASELINE configs[3] = 8 ranks x c3, configs[4] = 8 ranks x c5; 256-slide global
    batch, 5 stains) executes per step, on one GPU: its 32 local cases through the encoder, the replicated global InfoNCE over
    k_global <= 256 cases, and its share of GOT with the GLOBAL token count n = min(k_global, 256) (loss.py:282: indices are
    randperm(k) with k = the number of participating cases of the global batch) and supplied threshold extrema.  The other 7 ranks'
    contribution to the all-gathered payload is emulated (their presence labels drawn with the same rates; their slide embeddings =
    detached perturbed copies of the local ones); no collective runs.  Reports the step time and, against the single-rank step of the
    same configuration, the weak-scaling ceiling that the n = k_global GOT growth alone implies (communication not included).
    config "c3": dense 4096-patch bags, d = 512, ACROBAT presence rates -- or, with all_present (SURVEY 8(d)'s second mask variant,
    absent-bag rule wsi_dataset.py:66 never firing), every stain on every case: k_global = 256 -> n = 256, the largest GOT size class,
    four problems of 32 local cases each.  config "c5": ragged bags U[1024, 16384], d = 768, stain-encoding tokens, every stain
    present (as bench.py --config c5) -> the same four n = 256 problems on top of the ragged encoder."""
    B, M, N, Dm, _, stain_enc = CONFIGS[config]
    ragged = N == 0
    mods = MODS5[:M]
    torch.manual_seed(42)
    model = MADELEINE(make_cfg(M, Dm), stain_encoding=stain_enc).to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=FUSED_ADAMW)
    gen = torch.Generator(device=dev).manual_seed(1234)
    rates = torch.tensor([1.0, 0.46, 0.73, 0.73, 0.73])
    lab_g = (torch.rand(world * B, M, generator=torch.Generator().manual_seed(77)) < rates).float()
    if all_present or ragged:
        lab_g = torch.ones(world * B, M)
    lab_g[:, 0] = 1
    labels = lab_g[:B]
    if ragged:
        lens = torch.randint(1024, 16385, (B, M), generator=torch.Generator().manual_seed(4321))
        data = {"bags": [[torch.randn(int(lens[b, m]), Dm, device=dev, generator=gen) for m in range(M)] for b in range(B)],
                "modality_labels": labels}
        shape = f"ragged U[1024,16384] (mean {int(lens.float().mean())}) x {Dm} + stain tokens"
    else:
        feats = torch.randn(B, M, N, Dm, device=dev, generator=gen)
        feats = feats * labels.to(dev)[:, :, None, None]
        data = {"feats": feats, "modality_labels": labels}
        shape = f"{N} x {Dm}"
    crit = InfoNCE(temperature=0.001)
    largs = SimpleNamespace(global_loss="info-nce", symmetric_cl=True, local_loss_weight=1.0)
    k_g = [int(lab_g[:, s].sum()) for s in range(1, M)]
    noise = {m: 0.05 * torch.randn((world - 1) * B, 1, 512, device=dev, generator=gen) for m in mods}

    def rank_step(local_only=False):
        opt.zero_grad(set_to_none=True)
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=bf16):   # bf16: trainer.py:101-103 under `precision: bfloat16`
            embs, toks = model(data, device=dev)
            if local_only:   # the single-rank step of the same configuration (the weak-scaling reference: world size 1, n = k_local)
                loss, _ = D.calculate_losses_dp(mods[1:], crit, MF.HipGotImpl, embs, toks, labels[:, 1:], largs)
            else:
                loss = emulated_rank_loss(D, MF, crit, largs, mods, embs, toks, labels, lab_g, noise, world)
        loss.backward()
        opt.step()
        return loss

    el, loss, prof, _work, psteps = measure_leg(MF, rank_step, steps, warmup)
    got_ms = sum(prof[k][0] * prof[k][1] / psteps for k in ("got_fwd", "got_bwd", "got_bwd_finish") if k in prof)
    out = {"ms_per_step": round(1e3 * el / steps, 3), "steps": steps, "emulated_world": world,
           "workload": f"one rank of {world} x {config}{' (all stains present)' if all_present and not ragged else ''}"
                       f"{' under torch.autocast(bfloat16)' if bf16 else ''}: {B} local slides x {M} stains x "
                       f"{shape}; global batch {world * B} emulated: cases per stain {k_g} -> GOT token count n = min(k_global, 256) = "
                       f"{[min(k, 256) for k in k_g]}, replicated InfoNCE over k_global rows; no collective",
           "final_loss": float(loss.detach()), "got_ms_per_step_sum_over_stains": round(got_ms, 3),
           "kernel_ms": {n: round(v[0], 4) for n, v in prof.items()},
           "kernel_calls_per_step": {n: v[1] // psteps for n, v in prof.items()}}
    out["device_allocs_in_timed_region"] = measure_leg.last_allocs
    if ragged or all_present:
        # the single-rank step of the SAME data (the c3 leg serves the ACROBAT-mask variant): denominator of the weak-scaling ceiling
        el1, _l1, _p1, _w1, _ = measure_leg(MF, lambda: rank_step(True), steps, warmup, prof_steps=0)
        out["single_rank_ms_per_step"] = round(1e3 * el1 / steps, 3)
        out["implied_weak_scaling_ceiling_vs_single_rank"] = round(el1 / el, 4)
    return out


def secondary_inference_leg(dev, MF, MADELEINE, n_patches=30000, bags=20):
    """slide-embedding extraction as utils.run_inference drives it -- one full bag per call
    (batch 1, no gradients, nothing saved for backward) through encode_he."""
    torch.manual_seed(42)
    model = MADELEINE(make_cfg(2, 512)).to(dev).eval()
    gen = torch.Generator(device=dev).manual_seed(99)
    bag = torch.randn(1, n_patches, 512, device=dev, generator=gen)
    with torch.no_grad():
        for _ in range(3):
            model.encode_he(bag, dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(bags):
            model.encode_he(bag, dev)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        MF.TIMER = MF.KernelTimer()
        for _ in range(5):
            model.encode_he(bag, dev)
        prof = MF.TIMER.report()
        MF.TIMER = None
    out = {"value": round(bags / el, 2), "unit": "bags/s", "ms_per_bag": round(1e3 * el / bags, 3), "patches_per_bag": n_patches,
           "patches_per_sec": round(bags * n_patches / el), "workload": "encode_he, batch 1, fp32, no_grad",
           "kernel_ms": {n: round(v[0], 4) for n, v in prof.items()}}
    alg = n_patches * (4 * 512 * 4 + 4 * 4) + 4 * 512 * 4
    if "pool_fwd" in prof:
        out["pool_fwd_GBs"] = round(alg / (prof["pool_fwd"][0] * 1e-3) / 1e9, 1)
    # the reference's extraction script runs this loop under bf16 autocast (extract_slide_embeddings.py:49 -> utils.py:52-55)
    with torch.no_grad(), torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        for _ in range(3):
            model.encode_he(bag, dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(bags):
            model.encode_he(bag, dev)
        torch.cuda.synchronize()
        el16 = time.perf_counter() - t0
    out["bf16_autocast"] = {"value": round(bags / el16, 2), "unit": "bags/s", "ms_per_bag": round(1e3 * el16 / bags, 3),
                            "patches_per_sec": round(bags * n_patches / el16)}
    # several bags per launch set (utils.run_inference's default: MADELEINE.encode_he_bags, packed tokens + cu_seqlens; bit-identical to
    # one call per bag): the host's ~40 launches per call are shared by 4 bags and the pooling kernel sees 4 x 235 workgroups
    group = [bag[0] for _ in range(4)]
    with torch.no_grad():
        for _ in range(2):
            model.encode_he_bags(group, dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(bags // 4):
            model.encode_he_bags(group, dev)
        torch.cuda.synchronize()
        el4 = time.perf_counter() - t0
        MF.TIMER = MF.KernelTimer()
        for _ in range(3):
            model.encode_he_bags(group, dev)
        prof4 = MF.TIMER.report()
        MF.TIMER = None
    nb4 = 4 * (bags // 4)
    out["four_bags_per_launch"] = {"value": round(nb4 / el4, 2), "unit": "bags/s", "ms_per_bag": round(1e3 * el4 / nb4, 3),
                                   "patches_per_sec": round(nb4 * n_patches / el4),
                                   "kernel_ms_per_call": {n: round(v[0], 4) for n, v in prof4.items()}}
    if "pool_fwd" in prof4:
        out["four_bags_per_launch"]["pool_fwd_GBs"] = round(4 * alg / (prof4["pool_fwd"][0] * 1e-3) / 1e9, 1)
    # ... under bf16 autocast (the released checkpoint's `precision: bfloat16`) with run_inference(bags_per_launch=4); its bf16 default is 1
    # bag per launch set (reproducible across dataloader orders)
    with torch.no_grad(), torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        for _ in range(2):
            model.encode_he_bags(group, dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(bags // 4):
            model.encode_he_bags(group, dev)
        torch.cuda.synchronize()
        el416 = time.perf_counter() - t0
    out["four_bags_per_launch_bf16_autocast"] = {"value": round(nb4 / el416, 2), "unit": "bags/s", "ms_per_bag": round(1e3 * el416 / nb4, 3),
                                                 "patches_per_sec": round(nb4 * n_patches / el416)}
    return out


COMPACT_MAX = 4096   # bytes: the driver's record keeps the last ~8 KB of stdout and parses the LAST line -- round 5's 22-KB line was lost


def compact_line(out):
    """The ONE JSON line the driver parses (<= COMPACT_MAX bytes, printed LAST): the contract keys, `roofline` (A3 pooling forward),
    `cpu_baseline`, and <= 1 KB of headline per-kernel figures.  Everything else (secondary legs, variants, provenance notes) lives in
    the detail record (bench_detail.json + a `BENCH_DETAIL ` stdout line that does not start with `{`)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    c = {k: out[k] for k in keep}
    c["dtype"] = out["dtype_short"]
    cfg = out["config"]
    c["config"] = {k: cfg[k] for k in ("workload", "global_batch", "parallelism", "collective_backend", "ranks_seen", "grad_sync", "gemm_mode",
                                       "device_allocs_in_timed_region", "final_loss", "grad_sync_ms", "grad_sync_share_of_step",
                                       "host_numa_pin", "local_cases_per_stain") if k in cfg}
    if "roofline" in out:
        r = out["roofline"]
        c["roofline"] = {k: r[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                                           "avg_ms", "launches") if k in r}
    if "cpu_baseline" in out:
        b = out["cpu_baseline"]
        c["cpu_baseline"] = {k: b[k] for k in ("value", "unit", "cores", "kind", "sample", "gpu_over_cpu") if k in b}
        c["cpu_baseline"]["sample"] = str(c["cpu_baseline"].get("sample"))[:200]
    kr = out.get("kernel_roofline", {})
    if kr:   # [ms per call, fraction of the 8 TB/s HBM or dense-MFMA peak]
        # (+ MFMA-busy fraction of the kernel's cycles from the committed counter passes, for the matrix-core families)
        c["kernels"] = {k: [round(v["avg_ms"], 3), round(v["frac"], 3)] + ([round(v["mfma_busy"], 3)] if "mfma_busy" in v else [])
                        for k, v in kr.items()}
    if "roofline_mfma" in out:
        m = out["roofline_mfma"]
        c["roofline_mfma"] = {k: m[k] for k in ("bound", "achieved", "peak", "unit", "frac", "frac_of_sustained_peak_random_operands") if k in m}
    sec = {}
    for name in ("bf16_mode", "c3_mode", "c4_rank_emulation", "c4_rank_emulation_all_present", "c5_rank_emulation", "c5_rank_emulation_bf16",
                 "host_input_mode"):
        if name in out and "ms_per_step" in out[name]:
            sec[name] = out[name]["ms_per_step"]
    if sec:
        c["secondary_ms_per_step"] = sec
    c["detail"] = out.get("detail_path")
    line = json.dumps(c, separators=(",", ":"))
    for drop in ("secondary_ms_per_step", "roofline_mfma", "kernels"):   # never exceed the cap: shed the optional objects first
        if len(line) <= COMPACT_MAX:
            break
        c.pop(drop, None)
        line = json.dumps(c, separators=(",", ":"))
    assert len(line) <= COMPACT_MAX, len(line)
    return line


def emit(out):
    """Detail record first (file + a stdout line that does NOT start with `{`), the compact contract line LAST."""
    path = None
    for d in (os.path.join(ROOT, "gpurun_out"), ROOT, "/tmp"):
        try:
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, "bench_detail.json")
            with open(path, "w") as f:
                json.dump(out, f, indent=1)
            break
        except OSError:
            path = None
    out["detail_path"] = os.path.relpath(path, ROOT) if path and path.startswith(ROOT) else path
    print("BENCH_DETAIL " + json.dumps(out), flush=True)
    print(compact_line(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--eval-mode", action="store_true", help="dropout off (parity-mode timing; not the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-input", action="store_true",
                    help="batches start in host memory and go through the pinned double-buffered H2D stager "
                         "(PCIe-inclusive rate; NOT the headline value, which is measured on device-resident bags)")
    ap.add_argument("--cpu-sample", type=int, default=32,
                    help="slides in the CPU-oracle sample: default = the FULL 32-slide step (1 warm-up + 1 timed, ~45 s on 16 cores); "
                         "a smaller sample underestimates the CPU rate (4 slides: 1.0-1.1 slides/s against 1.46 for the full step)")
    ap.add_argument("--precision", default="float32", choices=["float32", "bfloat16"],
                    help="float32 = the parity path (headline value).  bfloat16 = the reference's `precision: bfloat16` "
                         "runs: forward + losses under torch.autocast, bf16 activation storage + bf16 MFMA in the kernels")
    ap.add_argument("--no-bf16-leg", action="store_true", help="skip the short secondary bf16-mode measurement")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 PMC passes behind roofline.traffic")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the secondary c3 (5 stains + GOT) and inference legs")
    ap.add_argument("--skip-absent", action="store_true",
                    help="SURVEY 8(f) N4: encode the all-zero bag of an absent stain once instead of once per case (c3/c4 "
                         "mask stains with ACROBAT's presence rates; no effect on c2).  Off by default: the reference encodes them all")
    ap.add_argument("--slides", type=int, default=None,
                    help="slides per rank instead of the configuration's (32): a REDUCED workload, named as such in config.workload -- for "
                         "debug runs of many gloo ranks on one GPU (tests); never the metric")
    ap.add_argument("--label-seed", type=int, default=77, help="seed of the ACROBAT stain-presence masks (rank r draws with seed + r)")
    a = ap.parse_args()

    if a.gpus > 1 and "RANK" not in os.environ:
        # turnkey N-GPU run: `python bench.py --gpus N ...` without a launcher re-executes itself as N ranks (one per GPU) under
        # torch.distributed.run on the loopback rendezvous -- the form the driver uses when IT launches (`python -m torch.distributed.run
        # --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`); rank 0 prints the one JSON line
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL's peer mappings need it on this driver
        env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // a.gpus)))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execvpe(cmd[0], cmd, env)

    from madeleine_amd import InfoNCE, MADELEINE
    from madeleine_amd import distributed as D
    from madeleine_amd import functional as MF

    rank, world, local_rank = D.init_from_env()
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher set WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU: the HIP kernels are the only backend"
    n_dev = torch.cuda.device_count()
    dev = torch.device("cuda", local_rank % n_dev)   # (several ranks per GPU only in gloo debug runs)
    torch.cuda.set_device(dev)
    numa = D.pin_to_gpu_numa(dev.index) if world > 1 else None   # one process per GPU: host threads next to the GPU's PCIe root

    B, M, N, Dm, use_got, stain_enc = CONFIGS[a.config]
    reduced = a.slides is not None and a.slides != B
    if a.slides is not None:
        B = a.slides
    mods = MODS5[:M]
    torch.manual_seed(42)
    model = MADELEINE(make_cfg(M, Dm), stain_encoding=stain_enc).to(dev)
    model.eval() if a.eval_mode else model.train()
    model.skip_absent_stains = bool(a.skip_absent)
    net = model
    dist_on = D.collectives_on()   # world > 1, or world == 1 launched by torch.distributed.run: the whole N-rank path (RCCL, DDP, gloo side group)
    gsync = None
    if dist_on:
        # gradient mean over the ranks: ONE flat 20-MB all-reduce after backward (distributed.FlatGradSync) -- the DDP wrapper's per-step
        # host bookkeeping measured +4.5 ms on this 24-ms step against +1.0 (tools/exp_ddp.py, profiles/r04_exp_ddp_*.txt);
        # BENCH_DDP=1 selects the wrapper (distributed.wrap_ddp: 8-MB buckets overlapped with backward)
        if os.environ.get("BENCH_DDP"):
            net = D.wrap_ddp(model, dev, use_local_loss=use_got)
        else:
            gsync = D.FlatGradSync(model, use_local_loss=use_got)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=FUSED_ADAMW)
    torch.manual_seed(1000 + rank)   # dropout seeds are drawn from torch's CPU generator: decorrelate the ranks
    crit = InfoNCE(temperature=0.001)
    got_impl = MF.HipGotImpl if use_got else None
    largs = SimpleNamespace(global_loss="info-nce", symmetric_cl=True, local_loss_weight=1.0)

    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    ragged = N == 0
    if ragged:
        lg = torch.Generator().manual_seed(4321 + rank)
        lens = torch.randint(1024, 16385, (B, M), generator=lg)
        N = int(lens.float().mean())
        bags = [[torch.randn(int(lens[b, m]), Dm, device=dev, generator=gen) for m in range(M)] for b in range(B)]
        feats = None
    else:
        feats = torch.randn(B, M, N, Dm, device=dev, generator=gen)
    labels = torch.ones(B, M)
    if M > 2 and not ragged:  # ACROBAT stain presence rates; absent stain -> all-zero bag (wsi_dataset.py:66)
        rates = torch.tensor([1.0, 0.46, 0.73, 0.73, 0.73][:M])
        labels = (torch.rand(B, M, generator=torch.Generator().manual_seed(a.label_seed + rank)) < rates).float()
        labels[:, 0] = 1
        feats = feats * labels.to(dev)[:, :, None, None]
    data = {"bags": bags, "modality_labels": labels} if ragged else {"feats": feats, "modality_labels": labels}

    host_iter = None
    if a.host_input and not ragged:
        from madeleine_amd.data import DevicePrefetcher

        def host_batches():
            pool = [torch.randn(B, M, N, Dm) for _ in range(3)]
            i = 0
            while True:
                yield {"feats": pool[i % 3], "modality_labels": labels}
                i += 1
        host_iter = iter(DevicePrefetcher(host_batches(), dev, depth=2))

    hgroup = D.host_group()   # gloo group for host-resident control data (None on one rank / when gloo is unavailable)

    def step(bf16=(a.precision == "bfloat16")):
        nonlocal data
        if host_iter is not None:
            data = next(host_iter)
        opt.zero_grad(set_to_none=True)
        # presence labels of the global batch: host-side exchange started now, waited for after the encoder forward has
        # been queued (no device work, no device synchronisation)
        pending = D.all_gather_labels_async(labels[:, 1:], hgroup)
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=bf16):   # as trainer.py:101-103
            embs, toks = net(data, device=dev)
            # ONE data-path collective: all-gather of [slide embeddings | presence mask | GOT extrema] -> replicated global
            # InfoNCE + rank-local GOT with global-batch thresholds ([S,6] all-reduce in backward); DDP all-reduces the grads.
            loss, flag = D.calculate_losses_dp(mods[1:], crit, got_impl, embs, toks, labels[:, 1:], largs,
                                               labels_global_withoutHE=pending.wait(), use_local_loss=use_got)
        loss.backward()
        if gsync is not None:
            if MF.TIMER is not None:       # profiled pass only: the packed copy + the one all-reduce between events on the launch stream
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                gsync.all_reduce_mean()
                e1.record()
                gsync_events.append((e0, e1))
            else:
                gsync.all_reduce_mean()
        opt.step()
        return loss

    gsync_events = []

    def fence():
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    # THE timed region: K steps, barrier + synchronize on both sides.  Only the A3 forward (the roofline kernel) carries HIP events here
    # -- one pair per step on the launch stream; the fused backward runs as its single call.  Every other per-kernel figure comes from
    # the short profiled pass below, outside the timed region.
    MF.TIMER = None
    MF.POOL_TIMER = None if os.environ.get("BENCH_NO_TIMER") else MF.PoolDispatchTimer()
    dev_allocs0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    dev_allocs = torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - dev_allocs0   # hipMalloc calls inside the timed region (0 = steady state)
    prof_pool, pool_parts = {}, None
    if MF.POOL_TIMER is not None:
        rep = MF.POOL_TIMER.report()          # the dispatches' own begin / end events (hipExtLaunchKernel), read after the region
        MF.POOL_TIMER = None
        if rep:
            prof_pool["pool_fwd"] = (sum(r[0] + r[1] for r in rep) / len(rep), len(rep))
            pool_parts = {"pool_partial_ms": round(sum(r[0] for r in rep) / len(rep), 4),
                          "pool_combine_ms": round(sum(r[1] for r in rep) / len(rep), 4),
                          "first_begin_to_last_end_ms": round(sum(r[2] for r in rep) / len(rep), 4)}
    # profiled pass (not part of `value`): every C-ABI launch between HIP events, the fused backward in its two phases
    prof, work, prof_steps, profiled_ms = {}, {}, max(3, min(5, a.steps)), None
    if not os.environ.get("BENCH_NO_TIMER"):
        MF.TIMER = MF.KernelTimer()
        tp = time.perf_counter()
        for _ in range(prof_steps):
            step()
        fence()
        profiled_ms = 1e3 * (time.perf_counter() - tp) / prof_steps
        prof, work = MF.TIMER.report(), MF.TIMER.work()
        MF.TIMER = None
        if "pool_fwd" in prof_pool:
            prof["pool_fwd"] = prof_pool["pool_fwd"]      # the roofline kernel: events of the timed region itself
    final_loss = float(loss.detach())
    bf16_leg = None
    if a.precision == "float32" and world == 1 and not a.no_bf16_leg and host_iter is None:
        # short secondary measurement of the bf16 mode (same model, same batch); reported beside, never as, `value`
        nb = max(3, min(10, a.steps))
        eb, lb, pb, wb, _ = measure_leg(MF, lambda: step(True), nb)
        bf16_leg = {"value": round(B * nb / eb, 3), "unit": "slides/s", "ms_per_step": round(1e3 * eb / nb, 3), "steps": nb,
                    "dtype": "bf16 activation storage + v_mfma_f32_32x32x16_bf16, fp32 accumulate/epilogues/params "
                             "(torch.autocast(bfloat16), the reference's `precision: bfloat16`)",
                    "final_loss": float(lb.detach()), "kernel_ms": {k: round(v[0], 4) for k, v in pb.items()},
                    "kernel_roofline": kernel_rooflines(pb, wb, "bfloat16")}
    f32_leg = None
    if a.precision == "float32" and world == 1 and not a.no_bf16_leg and host_iter is None and MF.gemm_mode() == "split":
        # the same step with every contraction on the exact-fp32 matrix-core kernels (v_mfma_f32_32x32x2_f32): the 'fp32' GEMM mode
        MF.set_gemm_mode("fp32")
        try:
            nf = max(3, min(10, a.steps))
            ef, lf, pf, wf, _ = measure_leg(MF, step, nf)
            f32_leg = {"value": round(B * nf / ef, 3), "unit": "slides/s", "ms_per_step": round(1e3 * ef / nf, 3), "steps": nf,
                       "dtype": "f32 on v_mfma_f32_32x32x2_f32 (MADELEINE_GEMM=fp32): the round-1/2 engine, 157.3 TFLOP/s peak",
                       "final_loss": float(lf.detach()), "kernel_ms": {k: round(v[0], 4) for k, v in pf.items()},
                       "kernel_roofline": kernel_rooflines(pf, wf, "float32", "fp32")}
        finally:
            MF.TIMER = None
            MF.set_gemm_mode("split")
    g2_leg = None
    if a.precision == "float32" and world == 1 and not a.no_bf16_leg and host_iter is None and MF.gemm_mode() == "split":
        # opt-in variant: backward products with two matrix terms (the non-gradient operand at its hi plane, functional.set_gradient_terms);
        # same forward bits, gradients at ~2^-12 relative.  Reported beside, never as, `value`.
        MF.set_gradient_terms(2)
        try:
            ng = max(3, min(10, a.steps))
            eg, lg, pg, wg, _ = measure_leg(MF, step, ng)
            g2_leg = {"value": round(B * ng / eg, 3), "unit": "slides/s", "ms_per_step": round(1e3 * eg / ng, 3), "steps": ng,
                      "dtype": "forward as the headline (3-term split-fp16); dX / dW products with 2 terms: weights (dX) and activations (dW) "
                               "at 11 bits, gradients ~2^-12 relative (tests/test_grad_terms_gpu.py)",
                      "final_loss": float(lg.detach()), "kernel_ms": {k: round(v[0], 4) for k, v in pg.items()}}
        finally:
            MF.TIMER = None
            MF.set_gradient_terms(3)
    # PCIe-inclusive leg ("a second number including pinned-host H2D"): the same step fed from HOST memory -- pinned
    # batches, as DataLoader(pin_memory=True) delivers them -- through data.DevicePrefetcher (side-stream H2D two batches ahead,
    # the step waits on the upload event only).  Never `value`.
    host_leg = None
    if a.precision == "float32" and world == 1 and not dist_on and not a.no_extra_legs and host_iter is None and not ragged:
        from madeleine_amd.data import DevicePrefetcher
        pool = [torch.randn(B, M, N, Dm).pin_memory() for _ in range(3)]

        def pinned_batches():
            i = 0
            while True:
                yield {"feats": pool[i % 3], "modality_labels": labels}
                i += 1
        it = iter(DevicePrefetcher(pinned_batches(), dev, depth=2))
        keep = data

        def host_step():
            nonlocal data
            data = next(it)
            return step()
        try:
            nh = max(5, min(10, a.steps))
            eh, lh, _ph, _wh, _ = measure_leg(MF, host_step, nh, warmup=3, prof_steps=0)
            mb = B * M * N * Dm * 4 / 2 ** 20
            host_leg = {"value": round(B * nh / eh, 3), "unit": "slides/s", "ms_per_step": round(1e3 * eh / nh, 3), "steps": nh,
                        "source": "pinned host batches (%.0f MiB of fp32 features per step) -> DevicePrefetcher: H2D on a side stream, depth 2; "
                                  "PCIe-inclusive, never `value`" % mb,
                        "h2d_GBs_needed_at_this_rate": round(mb * 2 ** 20 * nh / eh / 1e9, 1)}
        finally:
            it.close()
            data = keep
            del pool
    if host_iter is not None:
        host_iter.close()   # stops and joins the stager thread

    # socket power / shader clock over the same step loop (hwmon reads from a sampling thread; outside the timed region, after the
    # secondary legs of this model so that ~3 s at the power cap do not precede any of them)
    power = None
    if world == 1 and not dist_on and host_iter is None and not a.no_extra_legs:
        with PowerSampler() as ps:
            for _ in range(max(120, a.steps)):   # ~3 s: the hwmon averages update a few times per second
                step()
            fence()
        power = ps.summary(skip_s=1.0)
    c3_leg = infer_leg = c4_leg = c3ap_leg = c4ap_leg = c5_leg = c5b_leg = c3skip_leg = None
    if a.config == "c2" and a.precision == "float32" and world == 1 and not a.no_extra_legs and host_iter is None:
        # free the c2 working set first (the c3 step keeps ~60 GiB live)
        feats = data = None
        torch.cuda.empty_cache()
        c3_leg = secondary_c3_leg(dev, D, MF, InfoNCE, MADELEINE)
        torch.cuda.empty_cache()
        infer_leg = secondary_inference_leg(dev, MF, MADELEINE)
        torch.cuda.empty_cache()
        c4_leg = secondary_rank_leg(dev, D, MF, InfoNCE, MADELEINE)
        c4_leg["implied_weak_scaling_ceiling_vs_c3_single_rank"] = round(c3_leg["ms_per_step"] / c4_leg["ms_per_step"], 4)
        torch.cuda.empty_cache()
        # the all-present mask variant (every GOT problem in the n = 256 size class on a rank of config 4), and one rank of
        # config 5 (ragged, d = 768, stain tokens; every stain present -> the same four n = 256 problems)
        c3ap_leg = secondary_c3_leg(dev, D, MF, InfoNCE, MADELEINE, steps=4, all_present=True)
        torch.cuda.empty_cache()
        c3skip_leg = secondary_c3_leg(dev, D, MF, InfoNCE, MADELEINE, steps=4, skip_absent=True)
        torch.cuda.empty_cache()
        c4ap_leg = secondary_rank_leg(dev, D, MF, InfoNCE, MADELEINE, all_present=True)
        torch.cuda.empty_cache()
        c5_leg = secondary_rank_leg(dev, D, MF, InfoNCE, MADELEINE, config="c5", steps=3)
        torch.cuda.empty_cache()
        # ... and in the precision the reference runs that recipe in (scripts/launch_pretrain_withStainEncodings.sh: bf16 + stain tokens)
        c5b_leg = secondary_rank_leg(dev, D, MF, InfoNCE, MADELEINE, config="c5", steps=3, bf16=True)
        torch.cuda.empty_cache()

    tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if dist_on:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(tmax)
    ms_per_step = 1e3 * elapsed / a.steps
    value = B * world * a.steps / elapsed

    if rank == 0:
        tokens = int(lens.sum()) if ragged else B * M * N
        H = 4
        out = {
            "metric": "slides/sec (pretrain step) at B=32 N=4096 d=512; 1/2/4/8 GPU",
            "value": round(value, 3), "unit": "slides/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": (("f32 (values, accumulation and epilogues fp32; contractions as 3-term split-fp16 products ah bh + ah bl + al bh on "
                       "v_mfma_f32_32x32x16_f16 -- error below an fp32 fmaf chain, tests/test_split_gpu.py; MADELEINE_GEMM=fp32 selects "
                       "v_mfma_f32_32x32x2_f32, see fp32_mfma_mode)") if MF.gemm_mode() == "split" else "f32")
                      if a.precision == "float32" else "bf16 (storage + MFMA operands; fp32 accumulate)",
            "dtype_short": (("f32 (3-term split-fp16 MFMA products, fp32 accumulate)" if MF.gemm_mode() == "split" else "f32")
                            if a.precision == "float32" else "bf16 (storage + MFMA operands; fp32 accumulate)"),
            "data": ("synthetic (HOST-resident randn bags through the pinned double-buffered H2D stager, PCIe-inclusive)"
                     if host_iter is not None else
                     "synthetic (device-resident randn bags, random-init weights, manual_seed 42)"),
            "config": {"workload": f"{'REDUCED (--slides), not the metric: ' if reduced else ''}{a.config}: {B} slides/GPU x {M} stains x {'ragged U[1024,16384] (mean ' + str(N) + ')' if ragged else N} patches x {Dm}-d, "
                                   f"ABMIL pool + global InfoNCE{' + local GOT' if use_got else ''}, "
                                   f"{'eval (dropout off)' if a.eval_mode else 'train mode (dropout on)'}, AdamW"
                                   f"{', absent-stain zero bags encoded once (N4)' if a.skip_absent else ''}",
                       "global_batch": B * world, "bags_per_sec": round(value * M, 2), "parallelism": f"dp{world}",
                       "final_loss": final_loss, "optimizer": "torch.optim.AdamW(lr=1e-4%s)" % (", fused=True" if FUSED_ADAMW else ""),
                       "device_allocs_in_timed_region": int(dev_allocs),
                       "collective_backend": (torch.distributed.get_backend() if dist_on else "none"),
                       "ranks_seen": (torch.distributed.get_world_size() if dist_on else 1),
                       "host_label_exchange": ("gloo" if hgroup is not None else ("device" if dist_on else "none")),
                       "grad_sync": ("none" if not dist_on else ("ddp" if gsync is None else "flat_all_reduce"))},
        }
        if gsync_events:
            torch.cuda.synchronize()
            gms = sorted(e0.elapsed_time(e1) for e0, e1 in gsync_events)
            out["config"]["grad_sync_ms"] = round(gms[len(gms) // 2], 4)      # median over the profiled pass's steps
            out["config"]["grad_sync_share_of_step"] = round(gms[len(gms) // 2] / ms_per_step, 5)
        if numa is not None:
            out["config"]["host_numa_pin"] = numa
        if use_got and M > 2 and not ragged:
            out["config"]["local_cases_per_stain"] = [int(v) for v in labels[:, 1:].sum(0)]
        if "pool_fwd" in prof:
            ms, n = prof["pool_fwd"]
            esz = 4 if a.precision == "float32" else 2
            alg = tokens * (H * 512 * esz + H * 4) + B * M * H * 512 * 4
            ach = alg / (ms * 1e-3) / 1e9
            traffic, traffic_src = None, "not collected (only for c2 at N=1)"
            if a.config == "c2" and world == 1 and a.precision == "float32":
                if not a.no_pmc:
                    # the step pools from the split image of E in the split GEMM mode (same bytes): measure that instantiation
                    traffic, traffic_src = measure_pool_traffic(image=MF.gemm_mode() == "split")
                if traffic is None:
                    why = traffic_src
                    traffic = pool_traffic_from_profiles()
                    traffic_src = "committed PMC passes profiles/*_pool_pmc_{fetch,write}.txt (in-run collection: %s)" % why
            out["roofline"] = {"kernel": "abmil_pool_fwd (pool_partial + pool_combine)", "bound": "hbm",
                               "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                               "algorithmic_bytes_per_launch": alg, "avg_ms": round(ms, 4), "launches": n}
            if pool_parts is not None:
                out["roofline"].update(pool_parts)
                out["roofline"]["clock"] = ("avg_ms = pool_partial + pool_combine execution time, from the two dispatches' own start / stop events "
                                            "(hipExtLaunchKernel on the launch stream) inside the timed region -- the durations rocprofv3 "
                                            "--kernel-trace reports; the event-wrapped launches leave ~5 us between the two kernels "
                                            "(first_begin_to_last_end_ms) that the plain launches do not")
        if "gate_bwd_gemm" in prof and "gate_bwd" not in prof:   # the fused A2+A3 backward is timed in its two phases
            prof["gate_bwd"] = (prof["gate_bwd_gemm"][0] + prof["gate_bwd_dz"][0], prof["gate_bwd_gemm"][1])
        if "gate_fwd" in prof and "gate_bwd" in prof:
            msf, _ = prof["gate_fwd"]
            # MFMA-bound part of the backward: the dX / dW contractions (+ slab reduction).  The HBM-bound dz pass that precedes them
            # (reads the saved activations, writes dz) is timed separately and reported with its own HBM rate.
            split = "gate_bwd_gemm" in prof
            msb = prof["gate_bwd_gemm"][0] if split else prof["gate_bwd"][0]
            flop_f = tokens * H * 2 * 512 * 1024
            tf_f = flop_f / (msf * 1e-3) / 1e12
            tf_b = 2 * flop_f / (msb * 1e-3) / 1e12
            tf = 3 * flop_f / ((msf + msb) * 1e-3) / 1e12
            splitm = a.precision == "float32" and MF.gemm_mode() == "split"
            mpeak = (F16_MFMA_PEAK_TF if splitm else F32_MFMA_PEAK_TF) if a.precision == "float32" else 2500.0
            mult = 3.0 if splitm else 1.0   # fp16 MFMA FLOPs issued per algorithmic fp32 FLOP
            instr = "v_mfma_f32_32x32x16_bf16" if a.precision != "float32" else ("v_mfma_f32_32x32x16_f16 (3-term split products)" if splitm
                                                                                  else "v_mfma_f32_32x32x2_f32")
            out["roofline_mfma"] = {"kernel": "abmil_gate fwd + bwd contractions (dX, dW) on " + instr, "bound": "mfma",
                                    "achieved": round(mult * tf, 2), "peak": mpeak, "unit": "TFLOP/s",
                                    "frac": round(mult * tf / mpeak, 4),
                                    "frac_of_sustained_peak_random_operands": round(
                                        mult * tf / MFMA_SUSTAINED_RANDOM_TF["bf16" if a.precision != "float32" else ("f16" if splitm else "f32")], 4),
                                    "sustained_peak_note": "register-only MFMA stream on random operands under the 1400 W cap: fp16 1.66, bf16 1.82 "
                                                           "PFLOP/s, fp32 151.5 TFLOP/s (profiles/r04h_mfma_power_by_operand_content.txt); "
                                                           "`peak` / `frac` stay the guide's nominal dense peak",
                                    "algorithmic_fp32_tflops": round(tf, 2), "algorithmic_over_fp32_mfma_peak": round(tf / F32_MFMA_PEAK_TF, 4),
                                    "fwd_tflops": round(tf_f, 2),
                                    "bwd_contractions_tflops": round(tf_b, 2), "fwd_ms": round(msf, 3),
                                    "bwd_contractions_ms": round(msb, 3), "bwd_includes_dz_pass": not split,
                                    # the same FLOPs over forward + the WHOLE backward (dz pass included): comparable across rounds
                                    "fwd_plus_full_bwd_tflops": round(3 * flop_f / ((msf + prof["gate_bwd"][0]) * 1e-3) / 1e12, 2),
                                    "clock": "HIP events in the un-profiled timed region (profiles/ run 7-12 % slower: rocprofv3 "
                                             "lowers the clocks)"}
            if split:
                esz = 4 if a.precision == "float32" else 2
                msz = prof["gate_bwd_dz"][0]
                dz_bytes = tokens * H * 2048 * esz   # reads act_a, act_b [T,H,512] each, writes dz [T,H,1024]
                out["roofline_mfma"]["dz_pass"] = {"bound": "hbm", "ms": round(msz, 3), "algorithmic_bytes": dz_bytes,
                                                   "achieved_GBs": round(dz_bytes / (msz * 1e-3) / 1e9, 1),
                                                   "frac": round(dz_bytes / (msz * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        out["kernel_ms"] = {k: round(v[0], 4) for k, v in prof.items()}
        out["kernel_calls_per_step"] = {k: v[1] // (min(a.steps, 64) if k == "pool_fwd" and prof_pool else prof_steps) for k, v in prof.items()}
        out["kernel_ms_source"] = ("pool_fwd: the dispatches' start / stop HIP events inside the timed region; all other kernels: a separate profiled "
                                   "pass of %d steps after it (%s ms/step with per-launch events and the backward in two phases)"
                                   % (prof_steps, "n/a" if profiled_ms is None else "%.3f" % profiled_ms))
        if power is not None:
            out["power_clock"] = power
        out["kernel_roofline"] = kernel_rooflines(prof, work, a.precision, MF.gemm_mode())
        if a.precision == "float32" and MF.gemm_mode() == "split":
            for fam, busy in committed_mfma_busy().items():
                if fam in out["kernel_roofline"]:
                    out["kernel_roofline"][fam].update(busy)
        out["config"]["gemm_mode"] = MF.gemm_mode() if a.precision == "float32" else "bf16"
        if host_leg is not None:
            out["host_input_mode"] = host_leg
        if bf16_leg is not None:
            out["bf16_mode"] = bf16_leg
        if f32_leg is not None:
            out["fp32_mfma_mode"] = f32_leg
        if g2_leg is not None:
            out["grad_terms2_mode"] = g2_leg
        if c3_leg is not None:
            out["c3_mode"] = c3_leg
        if infer_leg is not None:
            out["inference_mode"] = infer_leg
        if c4_leg is not None:
            out["c4_rank_emulation"] = c4_leg
        if c3ap_leg is not None:
            out["c3_all_present"] = c3ap_leg
        if c3skip_leg is not None:
            out["c3_skip_absent_optin"] = c3skip_leg
        if c4ap_leg is not None:
            out["c4_rank_emulation_all_present"] = c4ap_leg
        if c5_leg is not None:
            out["c5_rank_emulation"] = c5_leg
        if c5b_leg is not None:
            out["c5_rank_emulation_bf16"] = c5b_leg
        if world == 1 and not a.no_cpu_baseline:
            try:
                nb = min(a.cpu_sample, B)
                out["cpu_baseline"] = cpu_baseline(nb, M, N, Dm, use_got, steps=1 if nb >= 16 else 2)
                out["cpu_baseline"]["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
                if a.config == "c2" and not a.no_extra_legs:
                    # SURVEY 8(d) context entries beside the train-mode C2 step (never the baseline `value` is compared with):
                    # the same step with dropout off (bounded sample) and the reference's own CPU-runnable config 1
                    extra = {}
                    try:
                        extra["c2_dropout_off"] = cpu_baseline(min(8, B), M, N, Dm, use_got, steps=1, train=False)
                        c1 = CONFIGS["c1"]
                        extra["c1_train"] = cpu_baseline(c1[0], c1[1], c1[2], c1[3], c1[4], steps=3, train=True)
                    except Exception as e:  # context only
                        extra["error"] = "%s: %s" % (type(e).__name__, e)
                    out["cpu_baseline"]["variants"] = extra
            except Exception as e:  # never lose the GPU line because the host box is short on RAM: fall back to a 4-slide sample
                try:
                    out["cpu_baseline"] = cpu_baseline(min(4, B), M, N, Dm, use_got)
                    out["cpu_baseline"]["sample"] += f" (the full-step sample failed: {type(e).__name__})"
                except Exception as e2:
                    out["cpu_baseline"] = {"value": None, "unit": "slides/s", "cores": usable_cores(), "kind": "port",
                                           "sample": f"failed: {type(e2).__name__}: {e2}"}
        emit(out)

    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
