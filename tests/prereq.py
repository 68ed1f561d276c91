"""Prerequisites of the encoder-level tests: binaries built from the reference's sources into oracle/_ref (git-ignored,
shipped to the GPU box with the tree). A missing one skips its tests, so that a checkout without the reference still
runs everything else - unless XMI_REQUIRE_REF=1, which turns every such skip into a failure: the setting for any run
whose green result is meant to include the encoder-level tests (`XMI_REQUIRE_REF=1 python -m pytest tests -m gpu`)."""
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
STRICT = os.environ.get("XMI_REQUIRE_REF", "") == "1"


def needs_ref(*names, how="make -C oracle enc trace"):
    """Decorator: the test needs oracle/_ref/<name> for every name given."""
    missing = [n for n in names if not os.path.exists(os.path.join(REFDIR, n))]
    reason = f"oracle/_ref/{', '.join(missing)} not built ({how})" if missing else ""
    if missing and STRICT:
        return pytest.mark.missing_prerequisite(reason)
    return pytest.mark.skipif(bool(missing), reason=reason)
