defmodule SendSlam.HipNif do
  @moduledoc """
  NIF stubs for libsendslam_orb.so (send-slam_amd/nif/sendslam_nif.c).  All five are dirty
  CPU-bound NIFs; they return `{:error, {code, message}}` and never raise.
  """
  @on_load :load
  def load, do: :erlang.load_nif(:filename.join(:code.priv_dir(:send_slam), ~c"sendslam_nif"), 0)
  def create(_device, _n_features), do: :erlang.nif_error(:nif_not_loaded)
  def set_calibration(_ref, _camera_id, _k8, _w, _h, _fps, _rgb), do: :erlang.nif_error(:nif_not_loaded)
  def extract(_ref, _camera_id, _pixels, _w, _h, _channels, _timestamp), do: :erlang.nif_error(:nif_not_loaded)
  def match(_ref, _query, _train, _th, _num, _den), do: :erlang.nif_error(:nif_not_loaded)
  def track(_ref, _camera_id, _pixels, _w, _h, _channels, _timestamp), do: :erlang.nif_error(:nif_not_loaded)
  def track_reset(_ref), do: :erlang.nif_error(:nif_not_loaded)
  # pipelined host-memory path (ss_pipe_*): lists of Mat binaries in, lists of per-frame results out
  def pipe_open(_device, _n_features, _w, _h, _channels, _batch, _depth, _match_mode, _rgb), do: :erlang.nif_error(:nif_not_loaded)
  def pipe_submit(_pipe, _frames, _camera_id, _first_timestamp, _dt), do: :erlang.nif_error(:nif_not_loaded)
  def pipe_wait(_pipe), do: :erlang.nif_error(:nif_not_loaded)
  # config 4: one HipBackend per eye; both call xchg_open with the same rendezvous path, then stereo_match once per stereo pair
  def xchg_open(_device, _rank, _world, _max_bytes, _rendezvous, _timeout_ms), do: :erlang.nif_error(:nif_not_loaded)
  def stereo_match(_ref, _xchg, _peer_rank), do: :erlang.nif_error(:nif_not_loaded)
end

defmodule SendSlam.HipBackend do
  @moduledoc """
  In-process replacement for the `SendSlam.DockerHandler` + `SendSlam.SlamHandler` pair on the
  ORB extract + match path.  Same GenServer call contract as DockerHandler
  (send_slam/lib/send_slam/docker_handler.ex:15-21,40-43) so `application.ex:83-96,117` needs a
  one-line child-spec swap; frames arrive exactly as SlamHandler receives them
  (`{:camera_frame, {:ok, opts}}`, slam_handler.ex:59) via `SendSlam.CameraRegistry`, and results
  leave as `{:broadcast_pose, map}` on `SendSlam.PoseRegistry` (slam_handler.ex:319-328).

  NOT COMPILED OR RUN in this repository's container (no BEAM); see INTEGRATION.md.
  """
  use GenServer
  require Logger

  @camera_registry SendSlam.CameraRegistry
  @calibration_registry SendSlam.CalibrationRegistry
  @pose_registry SendSlam.PoseRegistry

  def start_link(opts \\ []),
    do: GenServer.start_link(__MODULE__, opts, name: Keyword.get(opts, :server_name, __MODULE__))

  # DockerHandler's public API, same names, reply shapes and timeouts
  def start_container(pid), do: GenServer.call(pid, :start_container, 30_000)
  def stop_container(pid), do: GenServer.call(pid, :stop_container, 15_000)
  def status(pid), do: GenServer.call(pid, :status, 5_000)
  def logs(pid, lines \\ 100), do: GenServer.call(pid, {:logs, lines}, 15_000)

  # `auto_restart: true` (the option application.ex:94 passes and DockerHandler never reads): the
  # backend brings itself up in init/1, so a supervisor restart relaunches it -- with DockerHandler the
  # only :start_container call is the one in application.ex:117, made once at boot.
  @impl true
  def init(opts) do
    if Keyword.get(opts, :auto_restart, false), do: send(self(), :auto_start)

    {:ok,
     %{
       device: Keyword.get(opts, :device, 0),
       n_features: Keyword.get(opts, :n_features, 1250),
       ref: nil,
       state: :initial,
       container_id: nil,
       last_seen: nil,
       calibrated: false,
       log: :queue.new()
     }}
  end

  @impl true
  def handle_call(:start_container, _from, %{container_id: cid} = s) when is_binary(cid),
    do: {:reply, {:ok, cid}, s}

  def handle_call(:start_container, _from, s) do
    case SendSlam.HipNif.create(s.device, s.n_features) do
      {:ok, ref} ->
        {:ok, _} = Registry.register(@camera_registry, :clients, %{})
        {:ok, _} = Registry.register(@calibration_registry, :clients, %{})
        id = "hip-orb-gpu#{s.device}"
        {:reply, {:ok, id}, log(%{s | ref: ref, state: :running, container_id: id, last_seen: now_ms()}, "started #{id}")}

      {:error, reason} ->
        # like DockerHandler: stop so the supervisor restarts us; there is no CPU fallback
        {:stop, {:container_start_failed, reason}, {:error, reason}, %{s | state: :error}}
    end
  end

  def handle_call(:stop_container, _from, s), do: {:reply, :ok, %{s | ref: nil, state: :exited}}

  def handle_call(:status, _from, s),
    do: {:reply, %{state: s.state, container_id: s.container_id, last_seen: s.last_seen}, s}

  def handle_call({:logs, lines}, _from, s),
    do: {:reply, {:ok, s.log |> :queue.to_list() |> Enum.take(-lines) |> Enum.join("\n")}, s}

  @impl true
  def handle_info({:camera_frame, {:ok, opts}}, %{ref: ref} = s) when is_list(opts) and ref != nil do
    with {:ok, mat} <- Keyword.fetch(opts, :frame),
         {h, w, c} <- shape3(Evision.Mat.shape(mat)),
         {:ok, s} <- maybe_calibrate(s, opts, w, h) do
      camera_id = Keyword.get(opts, :camera_id, 1)
      ts = Keyword.get(opts, :timestamp, System.monotonic_time(:nanosecond) / 1.0e9)

      case SendSlam.HipNif.track(ref, camera_id, Evision.Mat.to_binary(mat), w, h, c, ts) do
        {:ok, 2 = state, {px, py, pz}, {qx, qy, qz, qw}, {n, _m, inliers, _pts}} ->
          # tracking state OK: the map SlamHandler forwards after unpacking a pose packet
          # (slam_handler.ex:114-127; built by the shim at orbslam3_mono_networked.cc:225-282)
          broadcast_pose(%{
            "type" => "pose",
            "timestamp" => ts,
            "camera_id" => camera_id,
            "tracking_state" => state,
            "position" => %{"x" => px, "y" => py, "z" => pz},
            "orientation" => %{"x" => qx, "y" => qy, "z" => qz, "w" => qw}
          })

          {:noreply, %{s | last_seen: now_ms()} |> log("frame #{camera_id}: #{n} keypoints, #{inliers} inliers")}

        {:ok, state, _pos, _quat, {n, _m, _i, _p}} ->
          # not OK: nothing is dispatched, exactly like the shim (orbslam3_mono_networked.cc:596)
          {:noreply, %{s | last_seen: now_ms()} |> log("frame #{camera_id}: #{n} keypoints, state #{state}")}

        {:error, reason} ->
          # bad frame => log + skip, never crash (orbslam3_mono_networked.cc:523-551)
          {:noreply, log(s, "frame skipped: #{inspect(reason)}")}
      end
    else
      _ -> {:noreply, s}
    end
  end

  def handle_info(:auto_start, %{container_id: nil} = s) do
    case handle_call(:start_container, nil, s) do
      {:reply, _reply, s2} -> {:noreply, s2}
      {:stop, reason, _reply, s2} -> {:stop, reason, s2}
    end
  end

  def handle_info(:auto_start, s), do: {:noreply, s}
  # A new calibration: the next frame calls set_calibration again, and ss_set_calibration resets the tracker (map,
  # reference frame, motion model) like the shim's rebuild of the whole System (orbslam3_mono_networked.cc:491-518).
  def handle_info({:broadcast_message, {:calibration, _calib}}, s), do: {:noreply, %{s | calibrated: false}}
  def handle_info(_other, s), do: {:noreply, s}

  # the string-keyed map of slam_handler.ex:127, to every process registered on PoseRegistry
  def broadcast_pose(pose_map) do
    Registry.dispatch(@pose_registry, :clients, fn entries ->
      for {pid, _} <- entries, do: send(pid, {:broadcast_pose, pose_map})
    end)
  end

  defp maybe_calibrate(%{calibrated: true} = s, _opts, _w, _h), do: {:ok, s}

  defp maybe_calibrate(s, opts, w, h) do
    case Keyword.get(opts, :calibration) do
      nil ->
        {:ok, s}

      calib ->
        [fx, _, cx, _, fy, cy | _] = calib[:camera_matrix] |> Evision.Mat.to_nx() |> Nx.to_flat_list()
        d = (calib[:distortion_coeffs] |> Evision.Mat.to_nx() |> Nx.to_flat_list()) ++ [0.0, 0.0, 0.0, 0.0]
        [k1, k2, p1, p2 | _] = d
        k8 = {fx * 1.0, fy * 1.0, cx * 1.0, cy * 1.0, k1 * 1.0, k2 * 1.0, p1 * 1.0, p2 * 1.0}
        fps = Keyword.get(opts, :fps, 30) * 1.0
        # rgb: 1 on BGR Mats, as slam_handler.ex:222 declares (SURVEY.md K0 note)
        case SendSlam.HipNif.set_calibration(s.ref, Keyword.get(opts, :camera_id, 1), k8, w, h, fps, 1) do
          :ok -> {:ok, %{s | calibrated: true}}
          {:error, reason} -> {:ok, log(s, "calibration rejected: #{inspect(reason)}")}
        end
    end
  end

  defp shape3({h, w, c}), do: {h, w, c}
  defp shape3({h, w}), do: {h, w, 1}
  defp shape3(_), do: :error
  defp now_ms, do: System.monotonic_time(:millisecond)

  defp log(s, line) do
    q = :queue.in(line, s.log)
    q = if :queue.len(q) > 2000, do: :queue.drop(q), else: q
    %{s | log: q}
  end
end
