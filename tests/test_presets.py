"""pepper_amd.variant.SetParameters against the reference's own set_parameters outputs
(tests/golden/variant_presets.json, produced by importing /root/reference: make_golden_presets.py)."""
import json
import os
from types import SimpleNamespace

from pepper_amd.variant.SetParameters import CANDIDATE_KEYS, IMAGE_KEYS, set_parameters

PROFILES = ["ont_r9_guppy5_sup", "ont_r9_guppy4_hac", "ont_r10_q20", "hifi", "clr"]


def test_presets_equal_reference(golden_dir, capsys):
    golden = json.load(open(os.path.join(golden_dir, "variant_presets.json")))
    names = IMAGE_KEYS + CANDIDATE_KEYS
    assert len(golden) == 29
    for case, want in golden.items():
        parts = case.split("/")
        o = SimpleNamespace(sub_command=parts[1], use_hp_info=False, **{n: None for n in names}, **{p: False for p in PROFILES})
        o.skip_indels = False
        for p in parts[0].split("+"):
            setattr(o, p, True)
        if len(parts) > 2:                      # the "user values survive" case
            o.skip_indels, o.min_mapq, o.snp_p_value = True, 42, 0.77
        set_parameters(o)
        assert {n: getattr(o, n) for n in names} == want, case
    capsys.readouterr()
