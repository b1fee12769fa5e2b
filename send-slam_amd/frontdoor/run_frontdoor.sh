#!/bin/bash
# run_frontdoor.sh -- answers the four `docker` invocations SendSlam.DockerHandler makes
# (/root/reference/send_slam/lib/send_slam/docker_handler.ex:117-182) by managing a
# sendslam_frontdoor process instead of a container, so the reference GenServer works unchanged
# with `docker_bin: ".../run_frontdoor.sh"`:
#
#   run -d --rm --name NAME --network=host [-e K=V]... IMAGE   -> starts the front door, prints an id
#   inspect -f '{{.State.Running}}' NAME                        -> prints true | false
#   logs --tail N NAME                                          -> last N lines of its output
#   rm -f NAME                                                  -> stops it
#
# State lives in ${SENDSLAM_RUN_DIR:-/tmp/sendslam}/NAME.{pid,log}.
set -u
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
BIN="${SENDSLAM_FRONTDOOR_BIN:-$HERE/sendslam_frontdoor}"
RUN_DIR="${SENDSLAM_RUN_DIR:-/tmp/sendslam}"
mkdir -p "$RUN_DIR"

cmd="${1:-}"; shift || true
case "$cmd" in
  run)
    name="sendslam"; envs=()
    while [ $# -gt 0 ]; do
      case "$1" in
        -d|--rm) shift ;;
        --network=*) shift ;;
        --name) name="$2"; shift 2 ;;
        -e) envs+=("$2"); shift 2 ;;
        *) shift ;;   # the image name: there is only one backend here
      esac
    done
    if [ -f "$RUN_DIR/$name.pid" ] && kill -0 "$(cat "$RUN_DIR/$name.pid")" 2>/dev/null; then
      echo "docker: Error response from daemon: Conflict. The container name \"/$name\" is already in use." >&2
      exit 125
    fi
    [ -x "$BIN" ] || { echo "run_frontdoor.sh: $BIN is not built" >&2; exit 127; }
    ( for kv in "${envs[@]}"; do export "$kv"; done; exec setsid "$BIN" ) >"$RUN_DIR/$name.log" 2>&1 &
    echo $! >"$RUN_DIR/$name.pid"
    printf '%s-%s\n' "$name" "$!" | sha256sum | cut -c1-64
    ;;
  inspect)
    name="${*: -1}"
    if [ -f "$RUN_DIR/$name.pid" ] && kill -0 "$(cat "$RUN_DIR/$name.pid")" 2>/dev/null; then echo true; else echo false; fi
    ;;
  logs)
    n=100; name="${*: -1}"
    while [ $# -gt 1 ]; do case "$1" in --tail) n="$2"; shift 2 ;; *) shift ;; esac; done
    [ -f "$RUN_DIR/$name.log" ] || { echo "Error: No such container: $name" >&2; exit 1; }
    tail -n "$n" "$RUN_DIR/$name.log"
    ;;
  rm)
    name="${*: -1}"
    if [ -f "$RUN_DIR/$name.pid" ]; then
      pid="$(cat "$RUN_DIR/$name.pid")"
      kill "$pid" 2>/dev/null
      for _ in 1 2 3 4 5 6 7 8 9 10; do kill -0 "$pid" 2>/dev/null || break; sleep 0.2; done
      kill -9 "$pid" 2>/dev/null
      rm -f "$RUN_DIR/$name.pid"
    fi
    echo "$name"
    ;;
  *)
    echo "run_frontdoor.sh: unsupported docker subcommand '$cmd'" >&2; exit 64 ;;
esac
