"""Per-user scratch directories for lock files and the service's sockets (standard library only: the thin service client
imports this).

Nothing here has a counterpart in the reference (its per-sample processes share nothing but the file system).  The point of
this module is that a directory under a world-writable place (/tmp) is only used when it is provably ours: another local user
can create ``/tmp/snpgpu-<uid>`` first, and whoever owns the directory owns what we later trust in it (sockets, lock files).
"""
import os
import stat
import tempfile


class UnsafeDirectory(OSError):
    """The directory exists but is not a private directory of this user."""


def ensure_private_dir(path, allow_group_read=False):
    """Create `path` (mode 0700, parents with default permissions) when missing, then insist that it is a real directory —
    not a symbolic link — owned by this user and closed to group and others (``allow_group_read``: only group/other WRITE is
    refused, for a directory the user named explicitly).  Returns the path; raises UnsafeDirectory otherwise."""
    try:
        os.makedirs(path, mode=0o700, exist_ok=True)
    except FileExistsError:                                     # a dangling link, or a file, where the directory should be
        pass
    try:
        st = os.lstat(path)
    except OSError as e:
        raise UnsafeDirectory("%s: %s" % (path, e))
    if not stat.S_ISDIR(st.st_mode):
        raise UnsafeDirectory("%s is not a directory (a symbolic link or a file is in its place)" % path)
    if st.st_uid != os.getuid():
        raise UnsafeDirectory("%s belongs to uid %d, not to this user" % (path, st.st_uid))
    closed_to = 0o022 if allow_group_read else 0o077
    if st.st_mode & closed_to:
        raise UnsafeDirectory("%s is open to other users (mode %o)" % (path, stat.S_IMODE(st.st_mode)))
    return path


def private_dir(*sub):
    """This user's private directory of the build, plus sub-directories (each checked the same way): ALWAYS
    ``<tmp>/snpgpu-<uid>`` — created 0700 and refused when somebody else got there first.  One place whatever the process was
    started from: rounds 3-4 preferred ``$XDG_RUNTIME_DIR``, so an interactive shell (which has one) and a scheduler job or a
    service started at boot (which have none) did not see each other's device-slot locks and service sockets
    (SNPGPU_MAX_PROCS_PER_DEVICE silently not enforced, a second service spawned).  Processes that are to share across users or
    with another temporary directory name a common one themselves: SNPGPU_LOCK_DIR (device slots, device.py) and
    SNPGPU_SERVICE=<dir> (sockets, service.py)."""
    path = ensure_private_dir(os.path.join(tempfile.gettempdir(), "snpgpu-%d" % os.getuid()))
    for name in sub:
        path = ensure_private_dir(os.path.join(path, name))
    return path
