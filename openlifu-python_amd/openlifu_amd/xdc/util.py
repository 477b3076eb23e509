"""``load_transducer_from_file`` (mirror of openlifu.xdc.util, xdc/util.py:10-30): one entry point for transducer JSON files of
either kind -- a ``Transducer`` or, marked by ``"type": "TransducerArray"``, a multi-module ``TransducerArray`` -- so that
callers (database, Slicer) need not know which they hold.  The flattened ``Transducer`` is what the HIP path uploads
(``Transducer.element_table``)."""
from __future__ import annotations

import json
import os
from typing import Union

from .transducer import Transducer
from .transducerarray import TransducerArray


def load_transducer_from_file(transducer_filepath: Union[str, os.PathLike], convert_array: bool = True):
    """Transducer for a transducer file; for a TransducerArray file the flattened Transducer (``to_transducer``,
    xdc/transducerarray.py:33-37) unless ``convert_array`` is False, then the TransducerArray itself.
    A missing file raises FileNotFoundError, as ``open`` does in the reference."""
    with open(transducer_filepath) as f:
        d = json.load(f)
    if d.get("type") == "TransducerArray":
        array = TransducerArray.from_dict(d)
        return array.to_transducer() if convert_array else array
    return Transducer.from_file(transducer_filepath)
