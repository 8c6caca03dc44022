"""hgr-net_amd: MI355X-native zero-shot forward path of HGR-Net.

Import as ``hgr_net_amd`` (the repo-root alias package points here).
"""
__version__ = "0.1.0"
