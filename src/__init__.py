"""Drop-in module paths: the names the reference's callers import (`src.prediction.models.dynamics`,
`src.cem.cem`, ...) resolved to the MI355X implementation in `robot_aware_control_amd`.

Every package of this shim extends its search path over the other `src` trees on `sys.path`
(`pkgutil.extend_path`), so with this repository AHEAD of the reference on PYTHONPATH the modules that exist here
shadow the reference's, and everything else (`src.dataset.wx250s.wx250s_model`, `src.utils.camera_calibration`,
`src.mbrl.*`, `src.env.*`) still resolves into the reference tree."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
