"""Put this directory first on PYTHONPATH to run the reference's inference.py unchanged on the MI355X path:

    PYTHONPATH=<repo>/dropin_site:<repo>:<reference> python -m torch.distributed.launch ... inference.py ...

(see panacea_amd/dropin.py and INTEGRATION.md)."""
try:
    import panacea_amd.dropin as _d
    import os as _os
    _d.install(lazy=True, first_stage=_os.environ.get("PANACEA_DROPIN_FIRST_STAGE") == "1",
               conditioner=_os.environ.get("PANACEA_DROPIN_CONDITIONER") == "1")
except Exception as _e:  # pragma: no cover - never break interpreter start-up
    import sys
    print(f"[panacea_amd] drop-in not armed: {_e}", file=sys.stderr)
