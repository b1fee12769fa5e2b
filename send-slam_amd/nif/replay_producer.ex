defmodule SendSlam.ReplayProducer do
  @moduledoc """
  Synthetic / replay frame source (SURVEY.md §8(f) rank 4): replays an image sequence
  (`frame_%06d.pgm` ...) into `SendSlam.CameraRegistry` with the event shape, pacing, looping and
  warm-up behaviour of `SendSlam.VideoProducer`'s image-sequence mode
  (send_slam/lib/send_slam/video_producer.ex:212-245, 303-357; usage sketch application.ex:60-72),
  so several cameras can be driven from files, one producer per `camera_id`.

  Differences by design: no reader process and no `VideoCapture` — one GenServer that reads the next
  file with `Evision.imread/2` on a `Process.send_after/3` tick, so a slow consumer never queues
  frames and the producer needs no reopen logic.

  NOT COMPILED OR RUN in this repository's container (no BEAM); its Python mirror
  `send-slam_amd/send_slam_amd/producer.py` is what the tests exercise.

      {SendSlam.ReplayProducer,
       [video_path: "/data/cam1/frame_%06d.pgm", fps: 30, loop: true, warmup_ms: 10_000, camera_id: 1]}
  """
  use GenServer
  require Logger

  @camera_registry SendSlam.CameraRegistry
  @calibration_registry SendSlam.CalibrationRegistry

  def start_link(opts), do: GenServer.start_link(__MODULE__, opts, name: Keyword.get(opts, :name, __MODULE__))

  @impl true
  def init(opts) do
    pattern = Keyword.fetch!(opts, :video_path)

    unless Regex.match?(~r/%0?\d*d/, pattern),
      do: raise(ArgumentError, "ReplayProducer replays image sequences: the path needs a %d field")

    first =
      Enum.find([0, 1], fn i -> File.exists?(sequence_filename(pattern, i)) end) ||
        raise ArgumentError, "no frame 0 or 1 for #{pattern}"

    {:ok, _} = Registry.register(@calibration_registry, :clients, %{})
    fps = Keyword.get(opts, :fps, 30)

    state = %{
      pattern: pattern,
      first: first,
      index: first,
      fps: fps,
      interval_ms: if(fps > 0, do: round(1000 / fps), else: 0),
      loop: Keyword.get(opts, :loop, false),
      warmup_ms: Keyword.get(opts, :warmup_ms, 0),
      warmup_until: nil,
      camera_id: Keyword.get(opts, :camera_id, 1),
      calibration: Keyword.get(opts, :calibration)
    }

    send(self(), :tick)
    {:ok, state}
  end

  @impl true
  def handle_info(:tick, state) do
    path = sequence_filename(state.pattern, state.index)

    cond do
      File.exists?(path) ->
        case Evision.imread(path, flags: Evision.Constant.cv_IMREAD_UNCHANGED()) do
          %Evision.Mat{} = mat ->
            broadcast(mat, state)
            schedule(state)
            {:noreply, advance(state)}

          other ->
            Logger.warning("ReplayProducer: cannot read #{path}: #{inspect(other)}")
            schedule(state)
            {:noreply, %{state | index: state.index + 1}}
        end

      state.loop and state.index != state.first ->
        send(self(), :tick)
        {:noreply, %{state | index: state.first, warmup_until: nil}}

      true ->
        {:stop, {:shutdown, :eof}, state}
    end
  end

  def handle_info({:broadcast_message, {:calibration, calib}}, state), do: {:noreply, %{state | calibration: calib}}
  def handle_info(_other, state), do: {:noreply, state}

  # the first frame is re-delivered until the warm-up time has passed, then the sequence moves on
  defp advance(%{index: i, first: f, warmup_ms: w} = state) when i == f and w > 0 do
    now = System.monotonic_time(:millisecond)
    until = state.warmup_until || now + w
    if now < until, do: %{state | warmup_until: until}, else: %{state | index: i + 1, warmup_until: nil}
  end

  defp advance(state), do: %{state | index: state.index + 1}

  defp schedule(%{interval_ms: ms}), do: Process.send_after(self(), :tick, max(ms, 1))

  defp broadcast(mat, state) do
    payload =
      {:ok,
       [
         frame: mat,
         calibration: state.calibration,
         timestamp: System.monotonic_time(:microsecond) / 1_000_000,
         fps: state.fps,
         camera_id: state.camera_id
       ]}

    Registry.dispatch(@camera_registry, :clients, fn entries ->
      for {pid, _} <- entries, do: send(pid, {:camera_frame, payload})
    end)
  end

  @doc "`frame_%06d.pgm`, 7 -> `frame_000007.pgm`"
  def sequence_filename(pattern, index) do
    Regex.replace(~r/%0?(\d*)d/, pattern, fn _, width ->
      w = if width == "", do: 0, else: String.to_integer(width)
      String.pad_leading(Integer.to_string(index), w, "0")
    end, global: false)
  end
end
