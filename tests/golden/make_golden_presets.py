"""Platform presets of the reference (pure Python, importable here): for every profile flag and sub-command, the
option values set_parameters fills in on an otherwise empty option set.
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_presets.py"""
import contextlib
import io
import json
import os
import sys
from types import SimpleNamespace

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
from pepper_variant.modules.argparse.SetParameters import set_parameters  # noqa: E402

NAMES = ["min_mapq", "min_snp_baseq", "min_indel_baseq", "snp_frequency", "insert_frequency", "delete_frequency",
         "min_coverage_threshold", "candidate_support_threshold", "snp_candidate_frequency_threshold",
         "indel_candidate_frequency_threshold", "skip_indels", "allowed_multiallelics", "snp_p_value", "insert_p_value",
         "delete_p_value", "snp_p_value_in_lc", "insert_p_value_in_lc", "delete_p_value_in_lc", "snp_q_cutoff",
         "indel_q_cutoff", "snp_q_cutoff_in_lc", "indel_q_cutoff_in_lc", "report_snp_above_freq", "report_indel_above_freq"]
PROFILES = ["ont_r9_guppy5_sup", "ont_r9_guppy4_hac", "ont_r10_q20", "hifi", "clr"]
out = {}
for profile in PROFILES + ["hifi+clr", "ont_r9_guppy5_sup+hifi"]:
    for sub in ("call_variant", "make_images", "find_candidates", "run_inference"):
        o = SimpleNamespace(sub_command=sub, use_hp_info=False, **{n: None for n in NAMES}, **{p: False for p in PROFILES})
        o.skip_indels = False
        for p in profile.split("+"):
            setattr(o, p, True)
        with contextlib.redirect_stderr(io.StringIO()):
            set_parameters(o)
        out[profile + "/" + sub] = {n: getattr(o, n) for n in NAMES}
# user-provided values survive
o = SimpleNamespace(sub_command="call_variant", use_hp_info=False, **{n: None for n in NAMES}, **{p: False for p in PROFILES})
o.skip_indels, o.hifi, o.min_mapq, o.snp_p_value = True, True, 42, 0.77
with contextlib.redirect_stderr(io.StringIO()):
    set_parameters(o)
out["hifi/call_variant/user_min_mapq_42_snp_p_0.77_skip_indels"] = {n: getattr(o, n) for n in NAMES}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "variant_presets.json")
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
print(len(out), "cases ->", path)
