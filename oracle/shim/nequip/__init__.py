"""ORACLE SHIM (test infrastructure, not product code).

Stand-in for the third-party `nequip` package (>=0.13.0, pyproject.toml:15-17 of the
reference), which is absent from this container and from /root/reference.  Only the
leaf names the reference imports exist (SURVEY.md §8c list).  Semantics restated from
memory of nequip's published behaviour -- PARITY UNPINNED
against nequip itself; random-weight parity between the HIP path and the reference
files is insensitive to these conventions because both share one state_dict.
"""
