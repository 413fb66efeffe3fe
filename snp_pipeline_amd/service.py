"""A per-node service behind the unchanged per-sample CLI.

run.py starts one ``cfsan_snp_pipeline call_consensus`` process per sample and flow (run.py:704-718) and one ``call_sites``
per sample (run.py:662-664).  In-process, such a call costs ~0.5 s of wall time around 70 ms of device work: interpreter and
numpy (0.13 s), HIP runtime + context (0.12 s), pinned staging buffers (0.05 s), and ~0.1 s of runtime tear-down at exit
(tools/cli_time.py).  With ``SNPGPU_SERVICE`` set, the console script becomes a thin client instead — standard library only,
no numpy, no HIP: it hands its argument list, working directory and environment to a long-lived server process over a unix
socket, prints what the server captured and exits with the server's exit code.  The server keeps the device context and the
pinned staging ring between requests and runs every request exactly as the CLI would (same parser, same exception hooks and
exit codes, same log text), one at a time per GPU.

    SNPGPU_SERVICE=auto     use the per-user default socket directory; start the server(s) when there are none
    SNPGPU_SERVICE=<dir>    an explicit directory (started on demand only with SNPGPU_SERVICE_SPAWN=1)
    unset / 0 / off         no service: everything in-process, as before

``cfsan_snp_pipeline serve`` starts the service by hand (one worker process per GPU; ``--device N`` one worker).  A client that
cannot reach a server does the work in-process.

Whom the client trusts: the socket directory must be a real directory of this user that nobody else can write to (the default
one is ``<tmp>/snpgpu-<uid>`` (the same with or without a login session), created 0700 and refused when another user made it first:
_paths.py), the process at the other end of the socket must run under this user's uid (SO_PEERCRED), and what it is sent of
the environment is the list of variables the steps read (``_forwarded``) — not the whole of ``os.environ``.
"""
from __future__ import print_function

import json
import os
import socket
import struct
import sys
import time

SERVED = ("call_sites", "filter_regions", "merge_sites", "call_consensus", "snp_matrix", "distance", "snp_reference")
_MAX_MESSAGE = 1 << 30


# What a step reads from its environment (the reference's configuration travels in environment variables: run.py exports the
# *_ExtraParams of snppipeline.conf and the error-handling switches), plus what the programs a step starts need.
_FORWARDED = frozenset((
    "errorOutputFile", "StopOnSampleError", "RemoveDuplicateReads", "EnableLocalRealignment", "CLASSPATH", "PATH", "JAVA_HOME",
    "PBS_JOBID", "JOB_ID", "SGE_TASK_ID", "SLURM_ARRAY_JOB_ID", "SLURM_JOBID", "SLURM_ARRAY_TASK_ID",
    "LANG", "LC_ALL", "LC_CTYPE", "TZ", "TMPDIR"))


def _forwarded(name):
    return name in _FORWARDED or name.endswith("_ExtraParams") or (name.startswith("SNPGPU_") and name != "SNPGPU_SERVICE")


def default_dir():
    """The per-user socket directory (created, and checked to be ours: _paths.private_dir raises UnsafeDirectory when it is not)."""
    from . import _paths
    return _paths.private_dir("service")


def _checked_dir(directory):
    """An explicitly named socket directory: it may be readable by others, but only this user may write to it."""
    from . import _paths
    return _paths.ensure_private_dir(directory, allow_group_read=True)


def _peer_is_me(conn):
    """True when the process that accepted this connection runs under our uid (SO_PEERCRED: pid, uid, gid of the peer)."""
    try:
        creds = conn.getsockopt(socket.SOL_SOCKET, socket.SO_PEERCRED, struct.calcsize("3i"))
        _, uid, _ = struct.unpack("3i", creds)
        return uid == os.getuid()
    except (OSError, AttributeError, struct.error):
        return False


def service_dir():
    """The socket directory the environment asks for, or None when the service is off."""
    v = os.environ.get("SNPGPU_SERVICE", "")
    if v in ("", "0", "off", "no", "false"):
        return None
    return default_dir() if v in ("auto", "1", "on", "yes", "true") else _checked_dir(v)


def _send(conn, obj):
    data = json.dumps(obj).encode("utf-8")
    conn.sendall(struct.pack("<Q", len(data)) + data)


def _recv_exact(conn, n):
    parts = []
    while n:
        b = conn.recv(min(n, 1 << 20))
        if not b:
            raise EOFError("connection closed")
        parts.append(b)
        n -= len(b)
    return b"".join(parts)


def _recv(conn):
    (n,) = struct.unpack("<Q", _recv_exact(conn, 8))
    if n > _MAX_MESSAGE:
        raise ValueError("message too long")
    return json.loads(_recv_exact(conn, n).decode("utf-8"))


def _sockets(directory):
    try:
        return sorted(os.path.join(directory, n) for n in os.listdir(directory) if n.startswith("dev") and n.endswith(".sock"))
    except OSError:
        return []


def _connect(path, timeout=None):
    s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    try:
        if timeout is not None:
            s.settimeout(timeout)
        s.connect(path)
        s.settimeout(None)
        return s
    except (OSError, socket.error):
        s.close()
        return None


# ---- client ----------------------------------------------------------------------------------------------------------------
def _spawn(directory):
    """Start the service (detached) unless somebody else is doing so, and wait until a socket answers."""
    import fcntl
    import subprocess
    _checked_dir(directory)
    with open(os.path.join(directory, "spawn.lock"), "a+") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)                     # one starter; the others wait here and then find the sockets
        try:
            if any(_probe(p) for p in _sockets(directory)):
                return
            script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bin", "cfsan_snp_pipeline")
            env = dict(os.environ)
            env.pop("SNPGPU_SERVICE", None)                  # the server itself works in-process
            with open(os.devnull, "r+b") as null, open(os.path.join(directory, "server.log"), "ab") as log:
                subprocess.Popen([sys.executable, script, "serve", "--socketDir", directory, "--idleTimeout", os.environ.get("SNPGPU_SERVICE_IDLE", "300")],
                                 stdin=null, stdout=log, stderr=log, start_new_session=True, env=env, cwd="/")
            deadline = time.time() + float(os.environ.get("SNPGPU_SERVICE_START_TIMEOUT", "60"))
            while time.time() < deadline:
                if os.path.exists(os.path.join(directory, "ready")) and any(_probe(p) for p in _sockets(directory)):
                    return
                time.sleep(0.02)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _probe(path):
    s = _connect(path, timeout=1.0)
    if s is None:
        return False
    try:
        _send(s, {"ping": True})
        return bool(_recv(s).get("pong"))
    except (OSError, EOFError, ValueError):
        return False
    finally:
        s.close()


def try_client(argv):
    """Run ``argv`` (the arguments after the program name) through the service.  Returns the exit code, or None when the
    request was not served (service off, subcommand not served, no server reachable): the caller then works in-process."""
    if not argv or argv[0] not in SERVED:
        return None
    if argv[0] == "call_sites":
        # Only the device pass belongs in the per-GPU worker.  With a VarScan jar on CLASSPATH (mode varscan) the step runs samtools
        # and a JVM, touches no device, and would queue behind the worker's one-request-at-a-time loop with its programs' stderr
        # going to the detached server instead of this command's: it runs in-process, as the reference runs it.
        try:
            from . import call_sites as _cs
            if _cs.site_calling_mode() != "device":
                return None
        except SystemExit:
            return None                                         # (a mode that does not exist: the in-process command reports it)
    try:
        directory = service_dir()
    except OSError as e:                                      # the directory is not provably ours: nothing in it is trusted
        sys.stderr.write("snpgpu service not used: %s\n" % e)
        return None
    if directory is None:
        return None
    may_spawn = os.environ.get("SNPGPU_SERVICE", "") in ("auto", "1", "on", "yes", "true") or os.environ.get("SNPGPU_SERVICE_SPAWN") == "1"

    def connect_any():
        # samples spread over the workers (one per GPU) by what they work on; a dead worker's neighbours take over
        socks = _sockets(directory)
        if not socks:
            return None
        start = sum(bytearray((os.getcwd() + "\0" + "\0".join(argv)).encode("utf-8", "surrogateescape"))) % len(socks)
        for k in range(len(socks)):
            c = _connect(socks[(start + k) % len(socks)])
            if c is not None:
                return c
        return None

    conn = connect_any()
    if conn is None and may_spawn:                           # no socket, or only the files a killed server left behind
        _spawn(directory)
        conn = connect_any()
    if conn is None:
        return None
    try:
        if not _peer_is_me(conn):
            sys.stderr.write("snpgpu service not used: the server at %s runs under another user\n" % directory)
            return None
        _send(conn, {"argv": list(argv), "argv0": sys.argv[0], "cwd": os.getcwd(),
                     "env": {k: v for k, v in os.environ.items() if _forwarded(k)}})
        reply = _recv(conn)
    except (OSError, EOFError, ValueError):
        return None                                          # (nothing has been printed yet: the in-process path starts clean)
    finally:
        conn.close()
    sys.stdout.write(reply.get("stdout", ""))
    sys.stdout.flush()
    sys.stderr.write(reply.get("stderr", ""))
    sys.stderr.flush()
    return int(reply.get("rc", 1))


# ---- server ----------------------------------------------------------------------------------------------------------------
def _run_request(req):
    """One CLI invocation inside the server process, with the client's argv / cwd / environment, output captured."""
    import io
    import traceback
    from . import cfsan_snp_pipeline as cli
    saved = (os.getcwd(), dict(os.environ), list(sys.argv), sys.stdout, sys.stderr, sys.excepthook)
    keep = {k: os.environ[k] for k in ("SNPGPU_DEVICE", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "HSA_ENABLE_IPC_MODE_LEGACY") if k in os.environ}
    out, err = io.StringIO(), io.StringIO()
    rc = 0
    try:
        os.chdir(req["cwd"])
        # the server's own environment, with the variables a step reads taken from the client — set, changed or absent as there
        for k in [k for k in os.environ if _forwarded(k)]:
            del os.environ[k]
        os.environ.update({k: v for k, v in req.get("env", {}).items() if _forwarded(k)})
        os.environ.update(keep)
        os.environ.pop("SNPGPU_SERVICE", None)
        sys.argv = [req.get("argv0", "cfsan_snp_pipeline")] + list(req["argv"])
        sys.stdout, sys.stderr = out, err
        args = None
        try:
            args = cli.parse_argument_list(list(req["argv"]))
            rc = cli.run_command_from_args(args)
        except SystemExit as e:
            rc = _exit_code(e, err)
        except BaseException:                                # noqa: B902 — what the process's excepthook would have got
            hook = getattr(args, "excepthook", None) or sys.__excepthook__
            try:
                hook(*sys.exc_info())
                rc = 1
            except SystemExit as e:
                rc = _exit_code(e, err)
            except BaseException:                            # noqa: B902
                traceback.print_exc(file=err)
                rc = 1
    finally:
        sys.stdout, sys.stderr, sys.excepthook = saved[3], saved[4], saved[5]
        sys.argv = saved[2]
        os.environ.clear()
        os.environ.update(saved[1])
        try:
            os.chdir(saved[0])
        except OSError:
            pass
    return {"rc": rc, "stdout": out.getvalue(), "stderr": err.getvalue()}


def _exit_code(e, err):
    if e.code is None:
        return 0
    if isinstance(e.code, int):
        return e.code
    err.write("%s\n" % (e.code,))
    return 1


def _worker(directory, device, idle_timeout):
    """Serve requests for one GPU until told to stop or idle for too long."""
    os.environ["SNPGPU_DEVICE"] = str(device)
    from . import _lib
    _lib.TORCH_FREE_OK = True
    from . import device as devmod
    dev = None
    try:
        dev = devmod.default_device()                        # the context, the staging ring and the code objects stay for all requests
    except Exception as e:                                   # noqa: B902 — the steps that need the device will say so themselves
        print("snpgpu service: no device %d (%s): only the host-side steps can be served" % (device, e))
    path = os.path.join(directory, "dev%d.sock" % device)
    if os.path.exists(path):
        if _probe(path):
            print("snpgpu service: device %d is already served at %s" % (device, path))
            return 0
        os.unlink(path)
    srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    old = os.umask(0o177)
    try:
        srv.bind(path)
    finally:
        os.umask(old)
    srv.listen(256)
    srv.settimeout(idle_timeout if idle_timeout > 0 else None)
    print("snpgpu service: device %d ready at %s" % (device, path))
    sys.stdout.flush()
    served = 0
    try:
        while True:
            try:
                conn, _ = srv.accept()
            except socket.timeout:
                print("snpgpu service: device %d idle for %g s, %d requests served: leaving" % (device, idle_timeout, served))
                break
            try:
                conn.settimeout(60.0)                        # a client that connects and says nothing does not hold the GPU's queue
                if not _peer_is_me(conn):                    # (the socket is 0600 in a 0700 directory; this is the second lock)
                    raise ValueError("connection from another user refused")
                req = _recv(conn)
                conn.settimeout(None)
                if req.get("ping"):
                    _send(conn, {"pong": True, "device": device, "served": served})
                elif req.get("stop"):
                    _send(conn, {"stopped": True})
                    break
                else:
                    _send(conn, _run_request(req))
                    served += 1
            except (OSError, EOFError, ValueError) as e:         # (socket.timeout is an OSError)
                print("snpgpu service: request dropped: %s" % e)
            finally:
                conn.close()
            sys.stdout.flush()
    finally:
        srv.close()
        try:
            os.unlink(path)
        except OSError:
            pass
        if dev is not None:
            dev.close()
    return 0


def serve(args):
    """Entry point of ``cfsan_snp_pipeline serve``."""
    import subprocess
    directory = _checked_dir(args.socketDir) if args.socketDir else (service_dir() or default_dir())
    if args.stop:
        for p in _sockets(directory):
            s = _connect(p, timeout=1.0)
            if s is not None:
                try:
                    _send(s, {"stop": True})
                    _recv(s)
                except (OSError, EOFError, ValueError):
                    pass
                finally:
                    s.close()
        return
    if args.device is not None:
        _worker(directory, int(args.device), float(args.idleTimeout))
        return
    from . import _lib
    _lib.TORCH_FREE_OK = True
    from . import device as devmod
    n = max(1, devmod.device_count())
    ready = os.path.join(directory, "ready")
    if os.path.exists(ready):
        os.unlink(ready)
    if n == 1:                                               # one GPU: this process is the worker
        with open(ready, "w") as f:
            f.write("1\n")
        _worker(directory, 0, float(args.idleTimeout))
        return
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bin", "cfsan_snp_pipeline")
    kids = [subprocess.Popen([sys.executable, script, "serve", "--socketDir", directory, "--idleTimeout", str(args.idleTimeout), "--device", str(i)])
            for i in range(n)]
    with open(ready, "w") as f:
        f.write("%d\n" % n)
    for k in kids:
        k.wait()


def add_arguments(sub):
    sub.add_argument("--socketDir", dest="socketDir", type=str, default=None, metavar="DIR", help="Directory of the service's unix sockets (default: $SNPGPU_SERVICE, or a per-user directory under the temporary directory)")
    sub.add_argument("--device", dest="device", type=int, default=None, metavar="INT", help="Serve this GPU only (default: one worker process per visible GPU)")
    sub.add_argument("--idleTimeout", dest="idleTimeout", type=float, default=0, metavar="SECONDS", help="Leave after this long without a request (0 = never)")
    sub.add_argument("--stop", dest="stop", action="store_true", help="Tell the running service to stop")
