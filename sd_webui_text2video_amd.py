"""Import shim: the package directory is named `sd-webui-text2video_amd/` (hyphens, as the task
prescribes), which Python cannot import by name.  This module makes it importable as
`sd_webui_text2video_amd` by turning itself into a package whose __path__ is that directory."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "sd-webui-text2video_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
