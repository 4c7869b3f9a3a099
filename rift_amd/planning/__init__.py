"""Host mirror of rift.cbv.planning (policy registry of the reference, rift/cbv/planning/__init__.py:21-34)."""


def __getattr__(name):
    if name == "CBV_POLICY_LIST":
        from rift_amd.planning.fine_tuner.rlft.rlft_pluto import CBV_POLICY_LIST
        return CBV_POLICY_LIST
    raise AttributeError(name)
