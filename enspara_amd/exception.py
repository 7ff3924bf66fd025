"""Exception types of the estimator surface.

Same names and meaning as the reference's (enspara/exception.py:5-40) so that
code written against enspara catches the same things.
"""


class ImproperlyConfigured(Exception):
    """Arguments that cannot be combined into a runnable configuration."""


class DataInvalid(Exception):
    """The data looks structurally invalid (mismatched lengths, wrong shape)."""


class InsufficientResourceError(Exception):
    """Valid data, but not enough memory / devices to carry out the request."""


class SuspiciousDataWarning(UserWarning):
    """Usable data with a suspicious structure or type."""


class PerformanceWarning(UserWarning):
    """Something happened that is likely to be slow and easy to avoid."""


class ConvergenceWarning(UserWarning):
    """An iterative procedure did not converge within its iteration budget."""
