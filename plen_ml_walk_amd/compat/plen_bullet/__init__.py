"""Import shim: put plen_ml_walk_amd/compat on PYTHONPATH and the reference's
`from plen_bullet import plen_env` (plen_bullet/src/plen_td3.py:7) resolves to the MI355X build."""
