"""Context options for the measurement tools: PRHF_TOOL_OPTIONS="name=value,..." is applied through
library.set_option (the library itself reads no environment variable)."""
import os


def apply(spec=None):
    spec = os.environ.get("PRHF_TOOL_OPTIONS", "") if spec is None else spec
    if not spec:
        return {}
    from pyrayhf_amd import library
    done = {}
    for item in spec.split(","):
        name, value = item.split("=")
        library.set_option(name.strip(), float(value))
        done[name.strip()] = float(value)
    return done
