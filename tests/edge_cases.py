"""Degenerate and extreme graphs, compared with the oracle (shared by the CPU-double and the
GPU edge-case tests)."""
import numpy as np
import pandas as pd


def directed_cases():
    rng = np.random.default_rng(5)
    star_in = pd.DataFrame({"from": np.arange(1, 900), "to": np.zeros(899, dtype=int)})
    star_out = pd.DataFrame({"from": np.zeros(899, dtype=int), "to": np.arange(1, 900)})
    both = pd.concat([star_in, star_out, pd.DataFrame({"from": [5, 6, 7], "to": [6, 7, 5]})],
                     ignore_index=True)
    n = 40
    a, b = np.nonzero(~np.eye(n, dtype=bool))
    complete = pd.DataFrame({"from": a, "to": b, "weight": rng.integers(1, 4, size=a.size)})
    chain = pd.DataFrame({"from": np.arange(0, 299), "to": np.arange(1, 300)})
    return {
        "single_self_loop": pd.DataFrame({"from": [7], "to": [7]}),
        "one_edge": pd.DataFrame({"from": [1], "to": [2]}),
        "two_cycle": pd.DataFrame({"from": [1, 2], "to": [2, 1]}),
        "star_in_900": star_in,            # one row with 899 entries, everything else empty
        "star_out_900": star_out,          # 899 rows gathering the same single row
        "star_both_ways": both,
        "complete_40": complete,           # dense -> MFMA GEMM legs in auto mode
        "chain_300": chain,
        "string_labels": pd.DataFrame({"from": list("abcdeab"), "to": list("bcdeaca")}),
    }


def bipartite_cases():
    rng = np.random.default_rng(6)
    one_item = pd.DataFrame({"user": np.arange(200), "item": np.zeros(200, dtype=int)})
    u = np.repeat(np.arange(30), 3)
    i = (np.arange(90) * 7) % 11
    key = np.unique(u * 100 + i)
    few = pd.DataFrame({"user": key // 100, "item": key % 100,
                        "weight": rng.integers(1, 6, size=key.size)})
    return {"one_item_200_users": one_item, "thirty_by_eleven": few,
            "single_pair": pd.DataFrame({"user": [3], "item": [9]})}
