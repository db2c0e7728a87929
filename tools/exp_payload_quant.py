"""VERDICT r2 #7 asked for a 32-byte voxel payload: six float32 information terms + the mean offset as 3 x 21-bit fixed point,
"H, b within 1e-5 of the UNQUANTISED oracle".  This measures that bar on the CPU oracle alone (no GPU): H, b at the guess and at
the true pose with the table payload in float64 (q0), float32 (q1: what the engine stores), 21-bit (q2) and 24-bit (q3) fixed
point over [-res/2, 3 res/2).  Result (profiles/r03_payload_quantisation.txt): b is 1.2e-4 ... 1.5e-4 off at the true pose with
21 bits and 1.5e-5 ... 1.9e-5 with 24 — over the bar — against 3e-6 ... 4e-6 for float32.  Three offsets that hold the bar
need >= 26 bits each, which leaves no room in 8 bytes: the 40-byte payload stays."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lio_slam_amd import synth
from oracle.oracle import Oracle


def rel(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


for idx, small in ((1, True), (1, False), (2, True)):
    c = synth.make_config(idx, small=small)
    O = {q: Oracle(quantize=q) for q in (0, 1, 2, 3)}
    for o in O.values():
        o.set_target(c["map"]); o.set_source(c["scan"])
    for name, T in (("guess", c["guess"]), ("true pose", c["T_true"])):
        e = {q: o.eval(T) for q, o in O.items()}
        print(f"configs[{idx}]{' small' if small else ''} at the {name}: n_eff " + " ".join(str(e[q]["n_effective"]) for q in e)
              + " | H rel vs float64: " + " ".join(f"q{q} {rel(e[q]['H'], e[0]['H']):.2e}" for q in (1, 2, 3))
              + " | b rel: " + " ".join(f"q{q} {rel(e[q]['b'], e[0]['b']):.2e}" for q in (1, 2, 3)))
