"""developer script (round 6): the fuzz shapes on which the one-launch kernel and the general path disagreed, solved by the
kernel alone (HIPSDP_SOLVE1_NO_FALLBACK=1), by the general path with its solves by substitution in the oracle's order (default since round 6)
and by the general path with the explicitly inverted factor (HIPSDP_SMALL_SOLVE=inverse, rounds 2-5), beside the oracle.
usage: python tests/devtools/small_solve_ab.py [seed ...]"""
import sys, os, importlib.util, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests', 'harness'), os.path.join(ROOT, 'tests')]
SEEDS = [int(a) for a in sys.argv[1:]] or [70071, 70081, 70387, 70505, 70602, 70680, 70733, 70740, 70951, 71064, 71150, 71191, 71239, 71248,
                                           50174, 50341, 50562, 50588, 30123, 30131, 30149]
if os.environ.get("SMALL_SOLVE_CHILD") is None:
    out = {}
    for mode in ("kernel", "subst", "inverse"):
        env = dict(os.environ, SMALL_SOLVE_CHILD=mode, HIPSDP_SOLVE1=("1" if mode == "kernel" else "0"), HIPSDP_SOLVE1_NO_FALLBACK="1")
        if mode == "inverse":
            env["HIPSDP_SMALL_SOLVE"] = "inverse"
        r = subprocess.run([sys.executable, __file__] + [str(s) for s in SEEDS], env=env, stdout=subprocess.PIPE, text=True)
        out[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    import ipm_ref
    from fuzz_shapes import problem
    agree_s = agree_i = 0
    for seed in SEEDS:
        core, desc = problem(seed)
        ref = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6, pabstol=1e-5))
        k, su, iv = out["kernel"][str(seed)], out["subst"][str(seed)], out["inverse"][str(seed)]
        agree_s += (k[0] == su[0]); agree_i += (k[0] == iv[0])
        print("seed %d %s: oracle (%d, %d) kernel (%d, %d) general/substitution (%d, %d) general/inverse (%d, %d)" % (
            seed, desc, ref.status, ref.iterations, k[0], k[1], su[0], su[1], iv[0], iv[1]))
    print("status agreement with the kernel: substitution %d of %d, inverse %d of %d" % (agree_s, len(SEEDS), agree_i, len(SEEDS)))
else:
    spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
    hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
    from fuzz_shapes import problem
    res = {}
    for seed in SEEDS:
        core, desc = problem(seed)
        s = hb.Solver(0); s.load_core(core)
        info = s.solve(gaptol=1e-6, feastol=1e-6, pabstol=1e-5)
        res[str(seed)] = (info.status, info.iterations, info.dobj, s.solve_path())
        s.close()
    print(json.dumps(res))
