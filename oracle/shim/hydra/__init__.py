"""ORACLE SHIM for hydra-core (absent): only `hydra.utils.instantiate` on `_target_` dicts."""
