"""Platform presets: fill the thresholds a user left unset (None) from the selected sequencing profile.

replaces: /root/reference/pepper_variant/modules/argparse/SetParameters.py:5-321 (`set_parameters`).  Same option
names, same precedence: `ont_r9_guppy5_sup` and `ont_r9_guppy4_hac` are applied if set (in that order), then the
first set flag of `ont_r10_q20`, `hifi`, `clr`; a value the user gave is never overwritten; image-generation
thresholds are touched for sub-commands call_variant / make_images / make_train_images, candidate-finding ones
for call_variant / find_candidates.  The tables below are the values the reference function produces on an empty
option set (tests/golden/variant_presets.json, written by tests/golden/make_golden_presets.py from the reference
itself; tests/test_presets.py compares every profile and sub-command).
"""
import sys
from datetime import datetime

IMAGE_KEYS = ['min_mapq', 'min_snp_baseq', 'min_indel_baseq', 'snp_frequency', 'insert_frequency', 'delete_frequency', 'min_coverage_threshold', 'candidate_support_threshold', 'snp_candidate_frequency_threshold', 'indel_candidate_frequency_threshold', 'skip_indels']

CANDIDATE_KEYS = ['allowed_multiallelics', 'snp_p_value', 'insert_p_value', 'delete_p_value', 'snp_q_cutoff', 'indel_q_cutoff', 'report_snp_above_freq', 'report_indel_above_freq', 'snp_p_value_in_lc', 'insert_p_value_in_lc', 'delete_p_value_in_lc', 'snp_q_cutoff_in_lc', 'indel_q_cutoff_in_lc']

PRESETS = {
    'ont_r9_guppy5_sup': {
        'image': {'min_mapq': 5, 'min_snp_baseq': 1, 'min_indel_baseq': 1, 'snp_frequency': 0.1, 'insert_frequency': 0.15, 'delete_frequency': 0.15, 'min_coverage_threshold': 3, 'candidate_support_threshold': 2, 'snp_candidate_frequency_threshold': 0.1, 'indel_candidate_frequency_threshold': 0.1, 'skip_indels': False},
        'candidate': {'allowed_multiallelics': 4, 'snp_p_value': 0.1, 'insert_p_value': 0.1, 'delete_p_value': 0.1, 'snp_q_cutoff': 20, 'indel_q_cutoff': 15, 'report_snp_above_freq': 0, 'report_indel_above_freq': 0, 'snp_p_value_in_lc': 0.1, 'insert_p_value_in_lc': 0.15, 'delete_p_value_in_lc': 0.1, 'snp_q_cutoff_in_lc': 20, 'indel_q_cutoff_in_lc': 10},
    },
    'ont_r9_guppy4_hac': {
        'image': {'min_mapq': 5, 'min_snp_baseq': 1, 'min_indel_baseq': 1, 'snp_frequency': 0.1, 'insert_frequency': 0.12, 'delete_frequency': 0.12, 'min_coverage_threshold': 3, 'candidate_support_threshold': 2, 'snp_candidate_frequency_threshold': 0.1, 'indel_candidate_frequency_threshold': 0.1, 'skip_indels': False},
        'candidate': {'allowed_multiallelics': 4, 'snp_p_value': 0.1, 'insert_p_value': 0.25, 'delete_p_value': 0.25, 'snp_q_cutoff': 20, 'indel_q_cutoff': 15, 'report_snp_above_freq': 0, 'report_indel_above_freq': 0, 'snp_p_value_in_lc': 0.05, 'insert_p_value_in_lc': 0.01, 'delete_p_value_in_lc': 0.01, 'snp_q_cutoff_in_lc': 20, 'indel_q_cutoff_in_lc': 10},
    },
    'ont_r10_q20': {
        'image': {'min_mapq': 1, 'min_snp_baseq': 1, 'min_indel_baseq': 1, 'snp_frequency': 0.1, 'insert_frequency': 0.1, 'delete_frequency': 0.1, 'min_coverage_threshold': 3, 'candidate_support_threshold': 2, 'snp_candidate_frequency_threshold': 0.1, 'indel_candidate_frequency_threshold': 0.1, 'skip_indels': False},
        'candidate': {'allowed_multiallelics': 4, 'snp_p_value': 1e-05, 'insert_p_value': 0.001, 'delete_p_value': 0.001, 'snp_q_cutoff': 15, 'indel_q_cutoff': 30, 'report_snp_above_freq': 0, 'report_indel_above_freq': 0, 'snp_p_value_in_lc': 1e-06, 'insert_p_value_in_lc': 0.001, 'delete_p_value_in_lc': 0.001, 'snp_q_cutoff_in_lc': 20, 'indel_q_cutoff_in_lc': 35},
    },
    'hifi': {
        'image': {'min_mapq': 5, 'min_snp_baseq': 10, 'min_indel_baseq': 10, 'snp_frequency': 0.1, 'insert_frequency': 0.12, 'delete_frequency': 0.1, 'min_coverage_threshold': 2, 'candidate_support_threshold': 2, 'snp_candidate_frequency_threshold': 0.1, 'indel_candidate_frequency_threshold': 0.1, 'skip_indels': False},
        'candidate': {'allowed_multiallelics': 4, 'snp_p_value': 0, 'insert_p_value': 0, 'delete_p_value': 0, 'snp_q_cutoff': 15, 'indel_q_cutoff': 20, 'report_snp_above_freq': 0, 'report_indel_above_freq': 0, 'snp_p_value_in_lc': 0, 'insert_p_value_in_lc': 0, 'delete_p_value_in_lc': 0, 'snp_q_cutoff_in_lc': 15, 'indel_q_cutoff_in_lc': 20},
    },
    'clr': {
        'image': {'min_mapq': 5, 'min_snp_baseq': 0, 'min_indel_baseq': 0, 'snp_frequency': 0.1, 'insert_frequency': 0.12, 'delete_frequency': 0.12, 'min_coverage_threshold': 3, 'candidate_support_threshold': 2, 'snp_candidate_frequency_threshold': 0.1, 'indel_candidate_frequency_threshold': 0.12, 'skip_indels': True},
        'candidate': {'allowed_multiallelics': 4, 'snp_p_value': 0.1, 'insert_p_value': 0.2, 'delete_p_value': 0.2, 'snp_q_cutoff': 20, 'indel_q_cutoff': 20, 'report_snp_above_freq': 0, 'report_indel_above_freq': 0, 'snp_p_value_in_lc': 0.05, 'insert_p_value_in_lc': 0.05, 'delete_p_value_in_lc': 0.05, 'snp_q_cutoff_in_lc': 20, 'indel_q_cutoff_in_lc': 20},
    },
}


def _apply(options, table, keys):
    for key in keys:
        if key == "skip_indels":
            # the reference tests `if not options.skip_indels` (False and None alike) and assigns the preset
            if not getattr(options, key, None):
                setattr(options, key, table[key])
        elif getattr(options, key, None) is None:
            setattr(options, key, table[key])


def set_parameters(options):
    sub = getattr(options, "sub_command", "call_variant")
    chosen = [p for p in ("ont_r9_guppy5_sup", "ont_r9_guppy4_hac") if getattr(options, p, False)]
    for p in ("ont_r10_q20", "hifi", "clr"):
        if getattr(options, p, False):
            chosen.append(p)
            break
    for p in chosen:
        if sub in ("call_variant", "make_images", "make_train_images"):
            _apply(options, PRESETS[p]["image"], IMAGE_KEYS)
        if sub in ("call_variant", "find_candidates"):
            _apply(options, PRESETS[p]["candidate"], CANDIDATE_KEYS)
    stamp = "[" + str(datetime.now().strftime('%m-%d-%Y %H:%M:%S')) + "] INFO: "
    sys.stderr.write(stamp + ("MODE: PEPPER HP\n" if getattr(options, "use_hp_info", False) else "MODE: PEPPER\n"))
    if chosen:
        sys.stderr.write(stamp + "PRESET: " + ", ".join(chosen) + "\n")
    return options
