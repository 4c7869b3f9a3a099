"""Host mirror of rift.cbv.planning (policy registry of the reference, rift/cbv/planning/__init__.py:21-34)."""


def __getattr__(name):
    if name == "CBV_POLICY_LIST":
        from rift_amd.planning.fine_tuner.rlft.rlft_pluto import CBV_POLICY_LIST
        if 'sft_pluto' not in CBV_POLICY_LIST:      # the SFT family (fine_tuner/sft/...): update side only, see sft_pluto.py
            from rift_amd.planning.fine_tuner.sft.rs_pluto.rs_pluto import RewardShapingPluto
            from rift_amd.planning.fine_tuner.sft.rtr_pluto.rtr_pluto import RTRPluto
            from rift_amd.planning.fine_tuner.sft.sft_pluto import SFTPluto
            CBV_POLICY_LIST.update({'sft_pluto': SFTPluto, 'rtr_pluto': RTRPluto, 'rs_pluto': RewardShapingPluto})
        return CBV_POLICY_LIST
    raise AttributeError(name)
