"""ORACLE SHIM: registry only (no compile machinery)."""
LMP_OUTPUTS = ["atomic_energy", "total_energy", "forces", "virial"]
COMPILE_TARGETS = {}


def single_frame_batch_map_settings(batch_map):
    return batch_map


def register_compile_targets(d):
    COMPILE_TARGETS.update(d)
