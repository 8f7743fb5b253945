"""Case tables shared by oracle/make_golden.py (generator) and tests/ (consumers).  TEST INFRASTRUCTURE ONLY."""

# (name, hparams name, weight seed, style, T, input seed)
MODEL_CASES = [
    ("small0_lively_T1500", "small0", 1, "lively", 1500, 3),
    ("small0_init_T1500", "small0", 1, "init", 1500, 3),
    ("small0_lively_T1012", "small0", 2, "lively", 1012, 4),
    ("small0_lively_T200", "small0", 2, "lively", 200, 5),
    ("final0_lively_T1500", "final0", 1, "lively", 1500, 3),
    ("final0_init_T1500", "final0", 1, "init", 1500, 3),
]
POSTP_CASES = {
    # name: list of (frame, value) spikes on a -5 floor, for both beat and downbeat rows
    "plateau2": ([(10, 2.0), (11, 2.0)], [(10, 1.0)]),
    "plateau3": ([(10, 2.0), (11, 2.0), (12, 2.0)], [(11, 1.0)]),
    "equal_two_apart": ([(20, 1.5), (22, 1.5)], [(20, 1.5), (22, 1.5)]),
    "lower_within3": ([(30, 3.0), (33, 2.0)], [(33, 2.0)]),
    "lower_at4": ([(30, 3.0), (34, 2.0)], [(34, 2.0)]),
    "zero_logit": ([(40, 0.0), (50, 1e-6)], [(50, 0.5)]),
    "equidistant": ([(60, 1.0), (70, 1.0)], [(65, 1.0)]),
    "no_beats": ([], [(15, 2.0)]),
    "edges": ([(0, 1.0), (99, 1.0)], [(0, 2.0), (99, 0.5)]),
    "dense": ([(i, 1.0 + 0.01 * (i % 7)) for i in range(3, 97, 2)], [(i, 1.0) for i in range(5, 95, 9)]),
}
