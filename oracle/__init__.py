

import os as _os
import sys as _sys

# (anything that imports this package may go on to import the reference from /root/reference: never write byte-code caches there,
#  neither from this process nor from its children)
_sys.dont_write_bytecode = True
_os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
