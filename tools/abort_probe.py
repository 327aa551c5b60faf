import os, sys, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.enable()
os.environ["PCT_ABORT_TRACE"] = "1"
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi
try:
    h = _capi.Handle(0)
except Exception as e:
    print("no device:", e)
os.abort()
