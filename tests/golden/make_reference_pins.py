"""Writes tests/golden/reference_pins.json: the known-answer vectors (inputs + expected outputs) that the reference's
own gtest files assert for the hot path, re-typed as data. Nothing is executed from /root/reference (it is C++ that
cannot be built offline); file:line of every vector is recorded next to it. reference.fasta / queries.fastq in this
directory are the reference's test data files (test/data/), copied verbatim as fixtures.
"""
import json
import os

REL = {"completely_above": 0, "completely_below": 1, "contains": 2, "equal": 3, "inside": 4,
       "overlapping_or_touching_above": 5, "overlapping_or_touching_below": 6}

iv = dict(ivl1=(5, 11), ivl2=(15, 21), ivl3=(11, 14), ivl4=(14, 15), ivl5=(0, 100), inside_ivl1=(6, 10),
          overlapping_below_ivl1=(3, 7), containing_ivl1=(3, 14), overlapping_below_ivl2=(13, 18),
          overlapping_above_ivl2=(17, 23), between_both=(11, 15), overlapping_both=(8, 16), containing_both=(3, 30),
          below_both=(0, 2), above_both=(22, 24))

others = ["inside_ivl1", "overlapping_below_ivl1", "containing_ivl1", "overlapping_below_ivl2", "overlapping_above_ivl2",
          "between_both", "overlapping_both", "containing_both", "below_both", "above_both"]

pins = {
    "math": {  # test/math_test.cpp:5-25
        "ceil_div": [[100, 8, 13], [100, 5, 20]],
        "fp_ceil": [[3.0, 3], [500 * 0.01, 5], [100 * 0.07, 7], [123.456, 124]],
        "saturate": [[42, 42], [2 ** 64 - 1, 2 ** 31 - 1]],
    },
    "input": {  # test/input_test.cpp:5-27
        "record_id": [["kcmieo25789377djs28 metadata", "kcmieo25789377djs28"]],
        "ranks": [["ACGTacgt", [1, 2, 3, 4, 1, 2, 3, 4]], ["ACGTacgt$", [1, 2, 3, 4, 1, 2, 3, 4, 0]],
                  ["ACGTacgtW3>", [1, 2, 3, 4, 1, 2, 3, 4, 5, 5, 5]]],
    },
    "pex": [  # test/pex_test.cpp:7-143 ; leaves as [from, to_inclusive, errors]
        {"len": 12, "k": 3, "s": 0, "bottom_up": False, "leaves": [[0, 2, 0], [3, 5, 0], [6, 8, 0], [9, 11, 0]]},
        {"len": 12, "k": 3, "s": 1, "bottom_up": False, "leaves": [[0, 5, 1], [6, 11, 1]]},
        {"len": 12, "k": 3, "s": 2, "bottom_up": False, "leaves": [[0, 5, 1], [6, 11, 1]]},
        {"len": 30, "k": 14, "s": 2, "bottom_up": True,
         "leaves": [[0, 5, 2], [6, 11, 2], [12, 17, 2], [18, 23, 2], [24, 29, 2]]},
    ],
    "erase_useless_anchors": {  # test/search_test.cpp:138-184 ; [pos, errors]
        "in": [[95, 5], [97, 3], [100, 0], [110, 10], [120, 0]], "out": [[100, 0], [120, 0]]},
    "search_seeds_setup": {  # test/search_test.cpp:6-75 (asserts only that nothing is fully excluded)
        "references": [[1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4], [1, 2, 3, 4, 1, 2, 3, 4]],
        "query": [1, 1, 1, 1, 1, 1, 2, 2, 2, 3, 2, 2, 1, 2, 3, 1, 2, 3, 4, 3, 2, 1, 4, 2],
        "seeds": [[0, 6, 0, 0], [6, 6, 1, 1], [12, 6, 1, 2], [18, 6, 0, 3]],  # offset, len, errors, leaf
        "config": {"hard": 10, "soft": 10, "order": "count_first", "choice": "round_robin", "erase": True},
    },
    "intervals": {  # test/intervals_test.cpp:35-157
        "named": {k: list(v) for k, v in iv.items()},
        "relationship": [
            ["ivl1", o, r] for o, r in zip(others + ["ivl1"], ["contains", "overlapping_or_touching_above", "inside",
                                                               "completely_below", "completely_below",
                                                               "overlapping_or_touching_below",
                                                               "overlapping_or_touching_below", "inside",
                                                               "completely_above", "completely_below", "equal"])
        ] + [
            ["ivl2", o, r] for o, r in zip(others + ["ivl2"], ["completely_above", "completely_above", "completely_above",
                                                               "overlapping_or_touching_above",
                                                               "overlapping_or_touching_below",
                                                               "overlapping_or_touching_above",
                                                               "overlapping_or_touching_above", "inside",
                                                               "completely_above", "completely_below", "equal"])
        ],
        "relationship_codes": REL,
        "trim": [[10, 20, 0, 10, 20], [10, 20, 1, 11, 19], [10, 20, 5, 14, 15], [10, 20, 10, 10, 11], [10, 20, 25, 10, 11]],
        "verified_intervals_steps": [  # insert list, then expected contains() for `others` in order
            {"insert": ["ivl1", "ivl2"], "contains_self": ["ivl1", "ivl2"],
             "expect": [True, False, False, False, False, False, False, False, False, False]},
            {"insert": ["ivl3"], "expect": [True, False, False, False, False, False, False, False, False, False]},
            {"insert": ["ivl4"], "expect": [True, False, False, False, False, False, False, False, False, False]},
            {"insert": ["ivl5"], "expect": [True] * 10},
        ],
        "others": others,
    },
    "alignment": {  # test/alignment_test.cpp:7-30
        "reference": [0, 0, 1, 2, 1, 3, 0, 2, 2, 3, 0, 1], "query": [1, 2, 1, 3, 1, 2, 2], "k": 2,
        "nm": 1, "start": 2, "cigar": "4=1X2="},
    "verification_verify": {  # test/verification_test.cpp:11-123
        "reference": [4, 2, 3, 4, 3, 4, 4, 4, 3, 2, 4, 3, 3, 2, 2, 3, 4, 4, 3, 3, 4, 3, 2, 2, 1, 4, 3, 3, 4, 2,
                      4, 4, 4, 3, 3, 2, 1, 1, 1, 2, 3, 4, 4, 3, 2, 4, 4, 2, 1, 4, 4, 3, 4, 4, 4, 4, 3, 3, 2, 1,
                      2, 3, 4, 3, 2, 1, 2, 3, 4, 3, 1, 4, 2, 1, 4, 4, 2, 2, 3, 4, 3, 3, 2, 1, 4, 4, 1, 1, 1, 2,
                      4, 3, 2, 1, 2, 2, 2, 3, 3, 1],
        "query": [4, 3, 4, 4, 4, 4, 3, 3, 2, 1, 4, 2, 3, 4, 3, 2, 1, 2, 3, 4, 1, 4, 2, 1, 4, 4, 2, 2, 3, 4],
        "k": 5, "s": 1, "bottom_up": True, "anchor": {"leaf": 0, "pos": 50, "errors": 0}, "extra_ratio": 0.1,
        "cigar": "10=1I9=1D10=", "nm": 2, "start": 50,
        "mutations_for_no_alignment": [[5, 1], [6, 1], [11, 3], [20, 2]]},
    "span": {  # test/verification_test.cpp:126-161
        "anchor_pos": 100755, "node": [500, 999, 30], "leaf_from": 750, "reflen": 1000000,
        "cases": [[0.0, [100475, 561, 0]], [0.01, [100469, 573, 6]]]},
    "try_align_node": {  # test/verification_test.cpp:163-261
        "reference": [2] * 10 + [1] * 80 + [2] * 10, "span": [50, 50], "node": [40, 84, 5],
        "query": [1, 1, 1, 3, 1, 1, 1, 1, 1, 1] + [1] * 30 + [1, 1, 1, 1, 1, 1, 1, 1, 1, 3, 1, 4, 1, 1, 1, 2, 1, 1, 1, 1,
                                                            1, 1, 1, 3, 1, 1, 1, 4, 1, 1] + [1] * 15,
        "nm": 5, "start": 50, "extra_error": [42, 2]},
    "whole_program": {  # test/floxer_whole_program_via_cli_test.cpp:17-143
        "args": {"query_errors": 2, "seed_errors": [0, 1], "extra_ratio": 2.0, "interval_opt": True},
        "unmapped": ["query1", "query6"],
        "expect": [  # id, reverse, pos_min, pos_max, nm, cigar
            ["query2", True, 48, 48, 0, "12="], ["query2", False, 11, 11, 0, "12="],
            ["query3", True, 17, 26, 2, "6=2I4="], ["query3", False, 36, 44, 2, "4=2I6="],
            ["query4", True, 7, 61, 2, "2I10="], ["query4", False, 54, 61, 2, "10=2I"],
            ["query5", True, 53, 53, 0, "12="], ["query5", False, 6, 6, 0, "12="]],
        "ids": ["query1", "query2", "query3", "query4", "query5", "query6"]},
}

here = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(here, "reference_pins.json"), "w") as f:
    json.dump(pins, f, indent=1)
print("written", len(json.dumps(pins)), "bytes")
