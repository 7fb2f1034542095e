"""Minimal logger: print, append to a file, or stay silent.

Same constructor and ``log(*items)`` call as the reference's
``visual_mpc/utils/logger.py:3-25`` (the CEM controller logs through it,
``cem_base_controller.py:21-26``).  The reference shells out to ``rm`` to clear an old log
file; here the file is truncated with ``os.remove``.
"""
import os


class Logger(object):
    def __init__(self, logfiledir=None, logfilename=None, printout=False, mute=False):
        self.logfiledir = logfiledir
        self.logfilename = logfilename
        self.mute = mute
        self.printout = printout or logfiledir is None or logfilename is None
        if logfiledir is not None and logfilename is not None:
            try:
                os.remove(os.path.join(logfiledir, logfilename))
            except OSError:
                pass

    def log(self, *items):
        if self.mute:
            return
        if self.printout:
            print(items)
            return
        with open(os.path.join(self.logfiledir, self.logfilename), 'a') as f:
            f.write(''.join(str(el) for el in items) + '\n')
