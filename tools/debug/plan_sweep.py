"""debug / verification sweep for the batch planner (GPU box).

The default planner evaluates most periods from claims proved for a bracket around the period's start
(gnsscorr_plan.hip); GNSSCORR_TRK_NOSPEC=1 selects the older chain that certifies every step itself, and
GNSSCORR_PLAN_VERIFY=1 keeps the claims but runs every step with its checks; GNSSCORR_TRK_ALGO=replica swaps the
prefix-sum correlator for the independent sample-by-sample form.  The four must agree bit for bit on everything a
batch returns.  This script runs one mode (argv[1]: default | nospec | verify | replica) over a seeded family of
configurations -- front ends (IQ at zero IF, real samples at 4.092 MHz IF), sampling rates, tap sets, Doppler up
to +-10 kHz, code frequency offsets up to +-12 chips/s, starts close to 0 and to 1 chip, GLONASS channels -- and
prints one digest line per configuration; `plan_sweep.py all` runs the four modes as child processes and compares
the lines."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NCFG = int(os.environ.get("SWEEP_NCFG", "48"))
NEPOCH = int(os.environ.get("SWEEP_NEPOCH", "400"))
NBATCH = 2


def configs():
    rng = np.random.default_rng(int(os.environ.get("SWEEP_SEED", "77001")))
    out = []
    for i in range(NCFG):
        kind = i % 6
        if kind in (0, 1, 2):
            fe = dict(dtype=2, f_if=0.0, f_sf=16.368e6)
        elif kind == 3:
            fe = dict(dtype=1, f_if=4.092e6, f_sf=16.368e6)
        elif kind == 4:
            fe = dict(dtype=2, f_if=0.0, f_sf=4.092e6)
        else:
            fe = dict(dtype=2, f_if=0.0, f_sf=20.0e6)
        taps = [(2, 3, 3), (1, 8, 8), (6, 3, 6), (2, 3, 3)][int(rng.integers(0, 4))]
        out.append(dict(fe=fe, taps=taps, seed=int(rng.integers(1, 1 << 30)), nch=int(rng.choice([7, 16, 32])),
                        dopp=float(rng.choice([500.0, 5000.0, 10000.0])), dcode=float(rng.choice([0.5, 3.0, 12.0]))))
    return out


def run_mode():
    sys.path.insert(0, ROOT)
    import ctypes as C
    import gnsscorr_loader
    gc = gnsscorr_loader.load()
    only = os.environ.get("SWEEP_ONLY")
    for ci, cfg in enumerate(configs()):
        if only is not None and ci != int(only):
            continue
        if ci < int(os.environ.get("SWEEP_FROM", "0")) or ci > int(os.environ.get("SWEEP_TO", "1000000")):
            continue
        eng = gc.Engine(0)
        fe, (corrn, corrd, corrp) = cfg["fe"], cfg["taps"]
        rng = np.random.default_rng(cfg["seed"])
        nsamp = int(fe["f_sf"] * 1e-3)
        nsamples = nsamp * (NEPOCH * NBATCH + 14)
        shape = (nsamples, 2) if fe["dtype"] == 2 else (nsamples,)
        data = rng.integers(-60, 61, size=shape, dtype=np.int8)
        eng.ring_create(1, fe["dtype"], nsamples)
        eng.ring_push_raw(1, data, nsamples)
        chans = [gc.Channel(1 + (p % 32), dtype=fe["dtype"], f_if=fe["f_if"], f_sf=fe["f_sf"], corrn=corrn, corrd=corrd,
                            corrp=corrp) for p in range(cfg["nch"])]
        eng.set_channels(chans)
        states = []
        for i, c in enumerate(chans):
            edge = i % 5
            remcode = (0.0 if edge == 0 else float(rng.uniform(0.0, 1e-6)) if edge == 1 else
                       float(1.0 - rng.uniform(0.0, 1e-6)) if edge == 2 else float(rng.uniform(0.01, 0.99)))
            states.append(dict(carrfreq=fe["f_if"] + float(rng.uniform(-cfg["dopp"], cfg["dopp"])),
                               codefreq=c.crate + float(rng.uniform(-cfg["dcode"], cfg["dcode"])),
                               remcode=remcode, remcarr=float(rng.uniform(0, 6.2831)) if i % 7 else 0.0,
                               buffloc=int(rng.integers(0, nsamp))))
        if cfg["nch"] > 3:
            states[3].update(carrfreq=fe["f_if"] + 200.0 * round(rng.uniform(-30, 30)), codefreq=chans[3].crate)   # acquisition grid
        eng.trk_set_state(states)
        stats = np.zeros(8, dtype=np.uint64)
        gc.lib().gnsscorr_debug_plan_stats(C.c_void_p(stats.ctypes.data), 1)
        h = hashlib.sha256()
        dump = []
        for b in range(NBATCH):
            eng.trk_run(NEPOCH)
            II, QQ, ns = eng.trk_fetch()
            if os.environ.get("SWEEP_DUMP") and ci == int(os.environ.get("SWEEP_DUMPCFG", "-1")):
                dump += [II.copy(), QQ.copy(), ns.copy()]
            h.update(np.ascontiguousarray(II).tobytes())
            h.update(np.ascontiguousarray(QQ).tobytes())
            h.update(np.ascontiguousarray(ns).tobytes())
        for f in eng.trk_get_state():
            h.update(np.array([f["remcode"], f["remcarr"]], dtype=np.float64).tobytes())
            h.update(np.array([f["buffloc"]], dtype=np.uint64).tobytes())
        gc.lib().gnsscorr_debug_plan_stats(C.c_void_p(stats.ctypes.data), 1)
        print(json.dumps(dict(cfg=ci, fe=fe, taps=cfg["taps"], nch=cfg["nch"], dopp=cfg["dopp"], dcode=cfg["dcode"],
                              digest=h.hexdigest(), stats=stats.tolist())), flush=True)
        if dump:
            np.savez(os.environ["SWEEP_DUMP"], *dump)
        eng.close()


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "all"
    if mode != "all":
        run_mode()
        return 0
    lines = {}
    for m, env_add in (("default", {}), ("nospec", dict(GNSSCORR_TRK_NOSPEC="1")), ("verify", dict(GNSSCORR_PLAN_VERIFY="1")),
                       ("replica", dict(GNSSCORR_TRK_ALGO="replica"))):
        env = dict(os.environ, **env_add)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), m], env=env, capture_output=True, text=True, timeout=int(os.environ.get("SWEEP_TIMEOUT", "240")))
        if out.returncode != 0:
            print(m, "failed:", out.stderr[-3000:])
            return 1
        lines[m] = [json.loads(x) for x in out.stdout.strip().splitlines() if x.startswith("{")]
        print(m, len(lines[m]), "configurations", flush=True)
    bad = 0
    tot = np.zeros(8, dtype=np.int64)
    for d, n, v, rp in zip(lines["default"], lines["nospec"], lines["verify"], lines["replica"]):
        same = d["digest"] == n["digest"] == v["digest"] == rp["digest"]
        tot += np.array(d["stats"], dtype=np.int64)
        if not same or v["stats"][6] != 0:
            bad += 1
            print("MISMATCH", d, n["digest"], v["digest"], rp["digest"], v["stats"])
    print("configurations", len(lines["default"]), "mismatches", bad, "planner paths (default mode, summed)", tot.tolist())
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
