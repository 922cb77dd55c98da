from gravomg.core import *   # noqa: F401,F403
from gravomg.util import *   # noqa: F401,F403
