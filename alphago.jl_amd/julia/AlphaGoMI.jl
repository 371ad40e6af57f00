#=
AlphaGoMI.jl -- thin `ccall` layer that puts libagz.so (MI355X / gfx950) behind the call surface of
tejank10/AlphaGo.jl's self-play hot path, so that `train()` / `selfplay` / the MCTSPlayer tests can
run against it unchanged (SURVEY.md 8b; ABI in include/agz.h).

NOTE: there is no `julia` binary in the build image, so this file has been written against the
Julia manual and the ABI header but never executed.  It contains no arithmetic of its own: every
function is argument marshalling (1-based <-> 0-based, column-major arrays passed as-is) around one
C call.  The Python mirror (alphago.jl_amd/api.py), which IS exercised by the test-suite, has the
same structure call for call.

Index conventions: the reference is 1-based -- flat move f in 1:N^2+1 with f == N^2+1 the pass,
coords (row, col) with row 1 at the top (src/game/go/coords.jl:5-12).  The ABI is 0-based:
a = f - 1, point p = (row-1) + N*(col-1).  Julia arrays are column-major, which is exactly the
layout the ABI documents for boards (N x N Int8), conv weights [kw,kh,cin,cout], dense [out,in],
features N x N x 17 x B and pi A x B -- so arrays cross without copies or permutes.
=#
module AlphaGoMI

using Printf: @sprintf
using Random

export GoEnv, Position, NeuralNet, MCTSPlayer, selfplay, extract_data, initialize_game!,
       tree_search!, pick_move, play_move!, should_resign, is_done, set_result!, all_legal_moves,
       score, result, result_string, IllegalMove, to_flat, from_flat, PlayerMove, BLACK, WHITE,
       SelfPlayPlayer, get_replay_batch, Momentum, _train, seed!,
       # the node-level surface test/test_mcts.jl:2-5 and test/test_mcts_player.jl:3-6 import
       MCTSNode, select_leaf, maybe_add_child!, add_virtual_loss!, revert_virtual_loss!,
       incorporate_results!, inject_noise!, child_action_score, child_Q, child_U, child_N, child_W,
       child_prior, N, Q, set_N!, suggest_move, get_position, children       # (`position` stays unexported: Base exports a function of that name; node.position and get_position serve)

const libagz = get(ENV, "AGZ_LIB", joinpath(@__DIR__, "..", "libagz.so"))

const BLACK, WHITE, EMPTY = 1, -1, 0

struct IllegalMove <: Exception end          # src/AlphaGo.jl:8

# ---- status codes (include/agz.h)
const AGZ_OK, AGZ_ILLEGAL_MOVE, AGZ_ASSERT_DONE_NODE, AGZ_HISTORY_INCOMPLETE, AGZ_BAD_SHAPE,
      AGZ_ASSERT_SOFTPICK = 0, 1, 2, 3, 4, 5
const AGZ_BAD_ARGUMENT, AGZ_HIP_ERROR, AGZ_POOL_EXHAUSTED, AGZ_RCCL_ERROR, AGZ_NOT_READY = 6, 7, 8, 9, 10

# agz_config: field order and types must match include/agz.h exactly (112 bytes)
struct AgzConfig
  board_size::Int32; tower_height::Int32; games::Int32; num_readouts::Int32
  parallel_readouts::Int32; two_player_mode::Int32
  komi::Float32; reserved0::Float32
  c_puct::Float64; dirichlet_noise_weight::Float64; resign_threshold::Float64
  resign_disable_fraction::Float64
  seed::UInt64; game_id_base::UInt64; game_id_stride::UInt64
  max_nodes_per_game::Int32; device::Int32; external_network::Int32; pool_policy::Int32
  record_capacity_games::Int32; arena_mode::Int32
end

struct AgzPositionInfo
  n::Int32; to_play::Int32; ko::Int32; caps_black::Int32; caps_white::Int32
  last_move::Int32; prev_move::Int32; history_len::Int32; komi::Float32
end

struct AgzNodeInfo
  N::Float32; W::Float32; Q::Float32
  parent::Int32; fmove::Int32; is_expanded::Int32; losses_applied::Int32; done::Int32
  pos::AgzPositionInfo
end

struct AgzGameHeader
  game_id::UInt64; num_moves::Int32; result::Int32; was_resign::Int32; resign_disabled::Int32
  final_score::Float32; short_searches::Int32
end

struct AgzStats            # agz_stats, include/agz.h: eighteen Int64 counters
  steps::Int64; positions::Int64; games_started::Int64; games_finished::Int64; evals::Int64
  duplicate_evals::Int64; terminal_visits::Int64; root_visits::Int64; nodes_in_use::Int64
  pool_exhausted::Int64; resigned_games::Int64; live_games::Int64; records_dropped::Int64
  pool_short_searches::Int64; peak_nodes_per_game::Int64; stalled_games::Int64; node_capacity::Int64
  abandoned_games::Int64
end

mutable struct Engine
  handle::Ptr{Cvoid}
  cfg::AgzConfig
end

function check(e::Engine, st::Integer)
  st == AGZ_OK && return
  msg = unsafe_string(ccall((:agz_last_error, libagz), Cstring, (Ptr{Cvoid},), e.handle))
  st == AGZ_ILLEGAL_MOVE && throw(IllegalMove())                       # board.jl:265,470
  st in (AGZ_ASSERT_DONE_NODE, AGZ_HISTORY_INCOMPLETE, AGZ_BAD_SHAPE, AGZ_ASSERT_SOFTPICK) &&
    throw(AssertionError(msg))                                          # mcts.jl:190,196 ...
  error("libagz status $st: $msg")
end

function stats(e::Engine)
  st = Ref{AgzStats}()
  check(e, ccall((:agz_engine_stats, libagz), Int32, (Ptr{Cvoid}, Ref{AgzStats}), e.handle, st))
  st[]
end

# a self-play game whose node pool is full plays its move early by default (agz_config.pool_policy, counted in
# stats(e).pool_short_searches and in the game's header); only a game that WAITS on its pool (AGZ_POOL_STALL, or the
# arena, which drops such a game) is an error here
check_pool(e::Engine) = (st = stats(e); (st.stalled_games > 0 || (e.cfg.arena_mode != 0 && st.pool_exhausted > 0)) &&
  error("libagz: a game is waiting on a full node pool (status $AGZ_POOL_EXHAUSTED); raise max_nodes_per_game"))

# agz_slot_status: per slot (status, nodes held, moves played); agz_slot_abandon drops the game in a slot
function slot_status(e::Engine)
  n = Int(e.cfg.games)
  st, nd, mv = zeros(Int32, n), zeros(Int32, n), zeros(Int32, n)
  check(e, ccall((:agz_slot_status, libagz), Int32, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}), e.handle, st, nd, mv))
  st, nd, mv
end
slot_abandon!(e::Engine, slot::Integer) =
  check(e, ccall((:agz_slot_abandon, libagz), Int32, (Ptr{Cvoid}, Int32), e.handle, slot))

function Engine(; board_size = 19, tower_height = 19, games = 1, num_readouts = 800,
                parallel_readouts = 8, two_player_mode = false, komi = 7.5, c_puct = 0.96,
                dirichlet_noise_weight = 0.25, resign_threshold = -0.9,
                resign_disable_fraction = 0.05, seed = 0, game_id_base = 0, game_id_stride = 1,
                max_nodes_per_game = 0, device = 0, external_network = false,
                record_capacity_games = 0, arena_mode = false, pool_policy = 0)
  cfg = AgzConfig(board_size, tower_height, games, num_readouts, parallel_readouts,
                  two_player_mode ? 1 : 0, komi, 0f0, c_puct, dirichlet_noise_weight,
                  resign_threshold, resign_disable_fraction, seed, game_id_base, game_id_stride,
                  max_nodes_per_game, device, external_network ? 1 : 0, pool_policy, record_capacity_games,
                  arena_mode ? 1 : 0)
  h = Ref{Ptr{Cvoid}}(C_NULL)
  st = ccall((:agz_engine_create, libagz), Int32, (Ref{AgzConfig}, Ref{Ptr{Cvoid}}), cfg, h)
  st == AGZ_OK || error("agz_engine_create: " *
    unsafe_string(ccall((:agz_last_error, libagz), Cstring, (Ptr{Cvoid},), C_NULL)))
  e = Engine(h[], cfg)
  finalizer(x -> ccall((:agz_engine_destroy, libagz), Cvoid, (Ptr{Cvoid},), x.handle), e)
  e
end

# ------------------------------------------------------------------ GoEnv / Position
# GoEnv(N, planes), src/game/go/go.jl:1-26
struct GoEnv
  N::Int
  action_space::Int
  planes::Int
  max_action_space::Int
  rules::Engine                 # a 1-slot engine used for the batched rule kernels with B = 1
end
GoEnv(board_size::Int = 19, planes::Int = 17) =
  GoEnv(board_size, board_size^2 + 1, (planes - 1) ÷ 2, 361,
        Engine(board_size = board_size, tower_height = 0, games = 1, num_readouts = 1,
               max_nodes_per_game = 8))

to_flat(c, env::GoEnv) = c === nothing ? env.N^2 + 1 : env.N * (c[2] - 1) + c[1]   # coords.jl:6-7
from_flat(f, env::GoEnv) = f == env.N^2 + 1 ? nothing : (1 + (f - 1) % env.N, 1 + (f - 1) ÷ env.N)

struct PlayerMove
  color::Int
  move::Union{Nothing, NTuple{2, Int}}
end

# GoPosition, src/game/go/board.jl:271-306 (the liberty tracker lives on the device side)
mutable struct Position
  env::GoEnv
  board::Matrix{Int8}
  n::Int
  komi::Float32
  caps::NTuple{2, Int}
  ko::Union{Nothing, NTuple{2, Int}}
  recent::Vector{PlayerMove}
  board_deltas::Array{Int8, 3}
  to_play::Int
  done::Bool
end
Position(env::GoEnv; board = zeros(Int8, env.N, env.N), n = 0, komi = 7.5, caps = (0, 0),
         ko = nothing, recent = PlayerMove[], board_deltas = zeros(Int8, env.N, env.N, 0),
         to_play = BLACK) =
  Position(env, board, n, komi, caps, ko, recent, board_deltas, to_play, false)

ko0(pos::Position) = pos.ko === nothing ? Int32(-1) : Int32(to_flat(pos.ko, pos.env) - 1)

# all_legal_moves(pos) -> Vector{Int8}(A), board.jl:393-424
function all_legal_moves(pos::Position)
  e = pos.env.rules
  out = zeros(Int8, pos.env.action_space)
  check(e, ccall((:agz_go_legal, libagz), Int32,
                 (Ptr{Cvoid}, Ptr{Int8}, Ptr{Int8}, Ptr{Int32}, Int32, Ptr{Int8}),
                 e.handle, pos.board, Int8[pos.to_play], Int32[ko0(pos)], 1, out))
  out
end

# score(pos), board.jl:511-533
function score(pos::Position)
  e = pos.env.rules
  out = zeros(Float32, 1)
  check(e, ccall((:agz_go_score, libagz), Int32, (Ptr{Cvoid}, Ptr{Int8}, Ptr{Float32}, Int32, Ptr{Float32}),
                 e.handle, pos.board, Float32[pos.komi], 1, out))
  out[1]
end
result(pos::Position) = (s = score(pos); s > 0 ? 1 : s < 0 ? -1 : 0)          # board.jl:535-544
result_string(s::Real) = s > 0 ? "B+" * @sprintf("%.1f", s) : s < 0 ? "W+" * @sprintf("%.1f", -s) : "DRAW"
result_string(pos::Position) = result_string(score(pos))                      # board.jl:546-555

# play_move!(pos, c; mutate = false), board.jl:451-509 / pass_move! :426-440
function play_move!(pos::Position, c; mutate = false)
  env = pos.env; e = env.rules; N = env.N
  a = to_flat(c, env) - 1
  newboard = similar(pos.board); ko = zeros(Int32, 1); ncap = zeros(Int32, 1); st = zeros(Int32, 1)
  check(e, ccall((:agz_go_play, libagz), Int32,
                 (Ptr{Cvoid}, Ptr{Int8}, Ptr{Int8}, Ptr{Int32}, Ptr{Int32}, Int32, Ptr{Int8}, Ptr{Int32},
                  Ptr{Int32}, Ptr{Int32}),
                 e.handle, pos.board, Int8[pos.to_play], Int32[ko0(pos)], Int32[a], 1, newboard, ko, ncap, st))
  st[1] == AGZ_ILLEGAL_MOVE && throw(IllegalMove())
  delta = newboard .- pos.board                       # +color at the move AND where stones vanished?
  # board.jl:479-481: delta[c] = color, delta[captured] = color  (so that prev = new - delta)
  delta = Int8.(ifelse.(delta .!= 0, Int8(pos.to_play), Int8(0)))
  caps = pos.to_play == BLACK ? (pos.caps[1] + ncap[1], pos.caps[2]) : (pos.caps[1], pos.caps[2] + ncap[1])
  keep = min(size(pos.board_deltas, 3), env.planes - 2)
  deltas = cat(dims = 3, reshape(delta, N, N, 1), pos.board_deltas[:, :, 1:keep])
  recent = vcat(pos.recent, PlayerMove(pos.to_play, c))
  done = c === nothing && !isempty(pos.recent) && pos.recent[end].move === nothing
  newko = ko[1] < 0 ? nothing : from_flat(ko[1] + 1, env)
  np = Position(env, newboard, pos.n + 1, pos.komi, caps, newko, recent, deltas, -pos.to_play, done)
  if mutate
    for f in fieldnames(Position); setfield!(pos, f, getfield(np, f)); end
    return pos
  end
  np
end
pass_move!(pos::Position; mutate = false) = play_move!(pos, nothing; mutate = mutate)

# ------------------------------------------------------------------ NeuralNet
# NeuralNet(env; tower_height), src/neural_net.jl:13-33.  Parameters are set layer by layer with
# agz_net_set_weights in Flux layout (conv [kw,kh,cin,cout], dense [out,in]); `load_flux!` walks a
# Flux Chain in `params` order (conv W,b ; BN beta,gamma ; ...), as save_model does (train.jl:14-35).
struct NeuralNet
  env::GoEnv
  tower_height::Int
  engine::Engine
end
function NeuralNet(env::GoEnv; tower_height::Int = 19, seed = 0)
  e = Engine(board_size = env.N, tower_height = tower_height, games = 1, num_readouts = 1,
             max_nodes_per_game = 8)
  check(e, ccall((:agz_net_init_synthetic, libagz), Int32, (Ptr{Cvoid}, UInt64), e.handle, seed))
  NeuralNet(env, tower_height, e)
end

set_weights!(nn::NeuralNet, layer::Integer, kind::Integer, data::AbstractArray{Float32}) =
  check(nn.engine, ccall((:agz_net_set_weights, libagz), Int32,
                         (Ptr{Cvoid}, Int32, Int32, Ptr{Float32}, Int64),
                         nn.engine.handle, layer, kind, data, length(data)))

# (nn)(positions::Vector{Position}) -> (pi A x B, v 1 x B), src/neural_net.jl:57-68
function (nn::NeuralNet)(positions::Vector{Position})
  env = nn.env; N = env.N; B = length(positions)
  boards = cat(dims = 3, (p.board for p in positions)...)
  deltas = zeros(Int8, N, N, 7, B)
  nd = zeros(Int32, B)
  for (b, p) in enumerate(positions)
    k = size(p.board_deltas, 3); nd[b] = k
    deltas[:, :, 1:k, b] .= p.board_deltas
  end
  tp = Int8[p.to_play for p in positions]
  pi = zeros(Float32, env.action_space, B); v = zeros(Float32, 1, B)
  check(nn.engine, ccall((:agz_net_forward, libagz), Int32,
        (Ptr{Cvoid}, Ptr{Int8}, Ptr{Int8}, Ptr{Int32}, Ptr{Int8}, Int32, Ptr{Float32}, Ptr{Float32}),
        nn.engine.handle, boards, deltas, nd, tp, B, pi, v))
  pi, v
end
(nn::NeuralNet)(pos::Position) = ((p, v) = nn([pos]); (p[:, 1], v[1]))        # neural_net.jl:70-73

# ------------------------------------------------------------------ MCTSPlayer (single tree, slot 0)
# MCTSPlayer(env, network; num_readouts, two_player_mode, resign_threshold), mcts_play.jl:3-24.
# `network` is any callable positions -> (pi, v) (duck-typed field, mcts_play.jl:5): a NeuralNet of
# this module is evaluated on the device without leaving it; anything else (DummyNet, a Flux model)
# receives the leaves' feature tensor through the select / incorporate split.
mutable struct MCTSPlayer
  env::GoEnv
  network
  num_readouts::Int
  two_player_mode::Bool
  τ_threshold::Int
  qs::Vector{Float32}
  searches_π::Vector{Vector{Float32}}
  result::Int
  result_string::String
  resign_threshold::Float64
  engine::Engine
  start::Union{Nothing, Position}       # the position initialize_game! was given
  recent::Vector{PlayerMove}            # start.recent + every move played since: root.position.recent (board.jl:299)
end
function MCTSPlayer(env::GoEnv, network; num_readouts = 800, two_player_mode = false,
                    resign_threshold = -0.9, seed = 0)
  τ = two_player_mode ? -1 : (env.N * env.N ÷ 12) ÷ 2 * 2
  th = network isa NeuralNet ? network.tower_height : 0
  e = Engine(board_size = env.N, tower_height = th, games = 1, num_readouts = num_readouts,
             parallel_readouts = 64, two_player_mode = two_player_mode,
             resign_threshold = resign_threshold, seed = seed, external_network = !(network isa NeuralNet))
  MCTSPlayer(env, network, num_readouts, two_player_mode, τ, Float32[], Vector{Float32}[], 0, "",
             resign_threshold, e, nothing, PlayerMove[])
end

function initialize_game!(p::MCTSPlayer, pos = nothing)                        # mcts_play.jl:110-118
  pos === nothing && (pos = Position(p.env))
  last = isempty(pos.recent) ? -1 : to_flat(pos.recent[end].move, p.env) - 1
  # the reference keeps pos.board_deltas (board.jl:505-506); the engine wants the older boards themselves,
  # newest first: B_{k+1} = B_k - delta_k (features.jl:8-14)
  k = min(7, size(pos.board_deltas, 3))
  hist = Matrix{Int8}(undef, length(pos.board), k)
  b = copy(pos.board)
  for i in 1:k
    b = b .- pos.board_deltas[:, :, i]
    hist[:, i] = vec(b)
  end
  prev = length(pos.recent) < 2 ? -1 : to_flat(pos.recent[end-1].move, p.env) - 1
  info = AgzPositionInfo(pos.n, pos.to_play, ko0(pos), pos.caps[1], pos.caps[2], last, prev, k, pos.komi)
  check(p.engine, ccall((:agz_tree_init, libagz), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Int8}, Ref{AgzPositionInfo}, Ptr{Int8}),
        p.engine.handle, 0, pos.board, info, k == 0 ? C_NULL : hist))      # ccall roots `hist` for the call
  p.result = 0; p.qs = Float32[]; p.searches_π = Vector{Float32}[]
  p.start = pos; p.recent = copy(pos.recent)
  p
end

function root(p::MCTSPlayer)
  r = Ref{Int32}(0)
  check(p.engine, ccall((:agz_tree_root, libagz), Int32, (Ptr{Cvoid}, Int32, Ref{Int32}), p.engine.handle, 0, r))
  r[]
end

function node_info(p::MCTSPlayer, node)
  info = Ref{AgzNodeInfo}()
  check(p.engine, ccall((:agz_tree_node_info, libagz), Int32, (Ptr{Cvoid}, Int32, Int32, Ref{AgzNodeInfo}),
                        p.engine.handle, 0, node, info))
  info[]
end
N(p::MCTSPlayer) = node_info(p, root(p)).N
Q(p::MCTSPlayer) = node_info(p, root(p)).Q

function child_N(p::MCTSPlayer)
  out = zeros(Float32, p.env.action_space)
  check(p.engine, ccall((:agz_tree_node_floats, libagz), Int32, (Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Float32}),
                        p.engine.handle, 0, root(p), 0, out))
  out
end

# The leaves the last agz_tree_search_select collected, as the reference's GoPosition fields (agz_tree_leaf_positions):
# node handles and, unless `nodes_only`, one Position per leaf -- board, board_deltas (newest first), to_play, n, ko,
# caps, the last two moves.
function leaf_positions(p::MCTSPlayer, B::Int; nodes_only::Bool = false)
  env = p.env; N = env.N
  nodes = zeros(Int32, B)
  if nodes_only
    check(p.engine, ccall((:agz_tree_leaf_positions, libagz), Int32,
          (Ptr{Cvoid}, Int32, Ptr{Int32}, Ptr{Int8}, Ptr{Int8}, Ptr{Int32}, Ptr{Int8}, Ptr{AgzPositionInfo}),
          p.engine.handle, 0, nodes, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL))
    return nodes, Position[]
  end
  boards = zeros(Int8, N, N, B); deltas = zeros(Int8, N, N, 7, B); nd = zeros(Int32, B); tp = zeros(Int8, B)
  info = Vector{AgzPositionInfo}(undef, B)
  check(p.engine, ccall((:agz_tree_leaf_positions, libagz), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Int32}, Ptr{Int8}, Ptr{Int8}, Ptr{Int32}, Ptr{Int8}, Ptr{AgzPositionInfo}),
        p.engine.handle, 0, nodes, boards, deltas, nd, tp, info))
  to_pm(a, color) = PlayerMove(color, a == N^2 ? nothing : from_flat(a + 1, env))
  positions = Position[]
  for b in 1:B
    i = info[b]
    recent = PlayerMove[]
    i.prev_move >= 0 && push!(recent, to_pm(Int(i.prev_move), Int(i.to_play)))
    i.last_move >= 0 && push!(recent, to_pm(Int(i.last_move), -Int(i.to_play)))
    push!(positions, Position(env, boards[:, :, b], Int(i.n), i.komi, (Int(i.caps_black), Int(i.caps_white)),
                              i.ko < 0 ? nothing : from_flat(i.ko + 1, env), recent,
                              deltas[:, :, 1:nd[b], b], Int(i.to_play), false))
  end
  nodes, positions
end

# Tracker-era networks answer with TrackedArrays (the reference's DummyNet returns `param(...)`,
# test/test_mcts_player.jl:22-32); mcts_play.jl:90 unwraps them with `.data`
untrack(x) = hasproperty(x, :data) ? getproperty(x, :data) : x

# tree_search!(player, parallel_readouts = 8), mcts_play.jl:73-98; returns the leaves (Vector{MCTSNode}) like the
# reference.  A caller-supplied network receives `positions::Vector{Position}` with length(positions) == number of
# leaves -- the reference's `mcts_player.network([leaf.position for leaf in leaves])` (mcts_play.jl:89).
function tree_search!(p::MCTSPlayer, parallel_readouts = 8)
  n = Ref{Int32}(0)
  check(p.engine, ccall((:agz_tree_search_select, libagz), Int32, (Ptr{Cvoid}, Int32, Int32, Ref{Int32}),
                        p.engine.handle, 0, parallel_readouts, n))
  B = Int(n[])
  if p.network isa NeuralNet || B == 0              # the engine's own network evaluates the leaves on the device
    nodes, _ = B == 0 ? (Int32[], Position[]) : leaf_positions(p, B; nodes_only = true)
    check(p.engine, ccall((:agz_tree_search_incorporate, libagz), Int32,
                          (Ptr{Cvoid}, Int32, Ptr{Float32}, Ptr{Float32}), p.engine.handle, 0, C_NULL, C_NULL))
    return [MCTSNode(p, id) for id in nodes]
  end
  nodes, positions = leaf_positions(p, B)
  move_probs, values = p.network(positions)                                     # mcts_play.jl:89
  move_probs, values = untrack(move_probs), untrack(values)                     # :90
  size(move_probs) == (p.env.action_space, B) && length(values) == B ||
    throw(AssertionError("network returned $(size(move_probs)) / $(length(values)) values for $B positions"))
  check(p.engine, ccall((:agz_tree_search_incorporate, libagz), Int32,
                        (Ptr{Cvoid}, Int32, Ptr{Float32}, Ptr{Float32}),
                        p.engine.handle, 0, Matrix{Float32}(move_probs), Vector{Float32}(vec(values))))
  [MCTSNode(p, id) for id in nodes]
end

function pick_move(p::MCTSPlayer)                                               # mcts_play.jl:52-71
  a = Ref{Int32}(0)
  check(p.engine, ccall((:agz_tree_pick_move, libagz), Int32, (Ptr{Cvoid}, Int32, Ref{Int32}), p.engine.handle, 0, a))
  from_flat(a[] + 1, p.env)
end

function play_move!(p::MCTSPlayer, c)                                           # mcts_play.jl:26-50
  info = node_info(p, root(p))
  π = child_N(p)
  if !p.two_player_mode
    # children_as_pi(root, n <= tau) (mcts.jl:241-252): the device records the same vector in its
    # game record; the host copy here serves `searches_π` for callers that read the field directly
    squash = info.pos.n <= p.τ_threshold
    pr = squash ? Float64.(π) .^ 0.98 : π
    push!(p.searches_π, Float32.(pr ./ sum(pr)))
  end
  push!(p.qs, info.Q)
  ok = Ref{Int32}(0)
  check(p.engine, ccall((:agz_tree_play_move, libagz), Int32, (Ptr{Cvoid}, Int32, Int32, Ref{Int32}),
                        p.engine.handle, 0, to_flat(c, p.env) - 1, ok))
  if ok[] == 0
    println("Illegal move")
    !p.two_player_mode && pop!(p.searches_π)
    pop!(p.qs)
    return false
  end
  push!(p.recent, PlayerMove(Int(info.pos.to_play), c))
  true
end

# ------------------------------------------------------------------ MCTSNode (node-level surface)
# The reference's tests drive single nodes (test/test_mcts.jl:2-5, test/test_mcts_player.jl:3-6): an MCTSNode here is
# a handle -- the player whose tree it lives in and the slot-local node id of include/agz.h -- and every function
# below is one agz_tree_* call.  `player.root` reads as in the reference (a property, below).
struct MCTSNode
  player::MCTSPlayer
  id::Int32
end
Base.getproperty(p::MCTSPlayer, s::Symbol) =
  s === :root ? MCTSNode(p, root(p)) : s === :position ? position(MCTSNode(p, root(p))) : getfield(p, s)
Base.:(==)(a::MCTSNode, b::MCTSNode) = getfield(a, :player) === getfield(b, :player) && getfield(a, :id) == getfield(b, :id)
# The reference's tests read a node's state as FIELDS (test/test_mcts.jl:52-70, test/test_mcts_player.jl:150-175:
# node.position, .children, .child_N, .child_prior, .fmove, .parent, .is_expanded ...; src/mcts.jl:41-53): each is one
# accessor call on the device tree (ADVICE r4).  `fmove` is 1-based as in the reference (nothing for a root); `parent`
# is an MCTSNode (nothing for a root: the reference's DummyNode carries no state a test reads).
function Base.getproperty(x::MCTSNode, s::Symbol)
  (s === :player || s === :id) && return getfield(x, s)
  s === :position && return position(x)
  s === :children && return children(x)
  s === :child_N && return child_N(x)
  s === :child_W && return child_W(x)
  s === :child_prior && return child_prior(x)
  s === :original_prior && return child_prior(x)      # mcts.jl:49: equal until inject_noise! (the device keeps one row)
  info = node_info(x)
  s === :fmove && return info.fmove < 0 ? nothing : Int(info.fmove) + 1
  s === :parent && return info.parent < 0 ? nothing : MCTSNode(getfield(x, :player), info.parent)
  s === :is_expanded && return info.is_expanded != 0
  s === :losses_applied && return Int(info.losses_applied)
  error("MCTSNode has no field $s")
end
Base.propertynames(::MCTSNode) = (:player, :id, :position, :children, :child_N, :child_W, :child_prior, :original_prior,
                                  :fmove, :parent, :is_expanded, :losses_applied)

node_info(x::MCTSNode) = node_info(x.player, x.id)
N(x::MCTSNode) = node_info(x).N                                                 # mcts.jl:98-100
Q(x::MCTSNode) = node_info(x).Q                                                 # mcts.jl:94
function node_floats(x::MCTSNode, field::Integer)
  e = x.player.engine
  out = zeros(Float32, x.player.env.action_space)
  check(e, ccall((:agz_tree_node_floats, libagz), Int32, (Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Float32}),
                 e.handle, 0, x.id, field, out))
  out
end
child_N(x::MCTSNode) = node_floats(x, 0)
child_W(x::MCTSNode) = node_floats(x, 1)
child_prior(x::MCTSNode) = node_floats(x, 2)
child_Q(x::MCTSNode) = child_W(x) ./ (1 .+ child_N(x))                          # mcts.jl:89
# mcts.jl:91-92, in the reference's own mixed precision (c_puct is Float64, the square root is Float32's)
child_U(x::MCTSNode) = (x.player.engine.cfg.c_puct * √(1 .+ N(x)) * child_prior(x) ./ (1 .+ child_N(x)))
function child_action_score(x::MCTSNode)                                         # mcts.jl:86-87 (Float64 scores)
  e = x.player.engine
  out = zeros(Float64, x.player.env.action_space)
  check(e, ccall((:agz_tree_node_scores, libagz), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{Float64}),
                 e.handle, 0, x.id, out))
  out
end
function set_N!(x::MCTSNode, value)                                             # mcts.jl:99
  e = x.player.engine
  check(e, ccall((:agz_tree_node_set_N, libagz), Int32, (Ptr{Cvoid}, Int32, Int32, Float32),
                 e.handle, 0, x.id, Float32(value)))
  value
end

function select_leaf(x::MCTSNode)                                               # mcts.jl:108-138
  e = x.player.engine
  leaf = Ref{Int32}(0)
  check(e, ccall((:agz_tree_select_leaf, libagz), Int32, (Ptr{Cvoid}, Int32, Int32, Ref{Int32}),
                 e.handle, 0, x.id, leaf))
  MCTSNode(x.player, leaf[])
end
function maybe_add_child!(x::MCTSNode, f::Integer)                              # mcts.jl:140-149; f is 1-based
  e = x.player.engine
  child = Ref{Int32}(0)
  check(e, ccall((:agz_tree_maybe_add_child, libagz), Int32, (Ptr{Cvoid}, Int32, Int32, Int32, Ref{Int32}),
                 e.handle, 0, x.id, f - 1, child))
  MCTSNode(x.player, child[])
end
# `up_to` is the node the walk towards the root stops at (mcts.jl:151-177); the tests pass `player.root`
function add_virtual_loss!(x::MCTSNode, up_to::MCTSNode)
  e = x.player.engine
  check(e, ccall((:agz_tree_add_virtual_loss, libagz), Int32, (Ptr{Cvoid}, Int32, Int32, Int32),
                 e.handle, 0, x.id, up_to.id))
end
function revert_virtual_loss!(x::MCTSNode, up_to::MCTSNode)
  e = x.player.engine
  check(e, ccall((:agz_tree_revert_virtual_loss, libagz), Int32, (Ptr{Cvoid}, Int32, Int32, Int32),
                 e.handle, 0, x.id, up_to.id))
end
function incorporate_results!(x::MCTSNode, move_probs, value, up_to::MCTSNode)  # mcts.jl:187-214
  e = x.player.engine
  probs = Float32.(vec(move_probs))
  check(e, ccall((:agz_tree_incorporate, libagz), Int32,
                 (Ptr{Cvoid}, Int32, Int32, Ptr{Float32}, Int32, Float32, Int32),
                 e.handle, 0, x.id, probs, length(probs), Float32(value), up_to.id))
end
function inject_noise!(x::MCTSNode)                                             # mcts.jl:232-239
  e = x.player.engine
  check(e, ccall((:agz_tree_inject_noise, libagz), Int32, (Ptr{Cvoid}, Int32, Int32), e.handle, 0, x.id))
end
function is_done(x::MCTSNode)                                                   # mcts.jl:227-229
  e = x.player.engine
  r = Ref{Int32}(0)
  check(e, ccall((:agz_tree_is_done, libagz), Int32, (Ptr{Cvoid}, Int32, Int32, Ref{Int32}), e.handle, 0, x.id, r))
  r[] != 0
end
function children(x::MCTSNode)                                                  # Dict(flat move => child), mcts.jl:51
  e = x.player.engine
  ids = zeros(Int32, x.player.env.action_space)
  check(e, ccall((:agz_tree_node_children, libagz), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{Int32}),
                 e.handle, 0, x.id, ids))
  Dict(f => MCTSNode(x.player, ids[f]) for f in eachindex(ids) if ids[f] >= 0)
end
function position(x::MCTSNode)                                                  # the node's GoPosition, mcts.jl:44
  p = x.player; e = p.engine; env = p.env
  info = node_info(x)
  board = zeros(Int8, env.N, env.N)
  check(e, ccall((:agz_tree_node_board, libagz), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{Int8}),
                 e.handle, 0, x.id, board))
  pos = Position(env)
  pos.board = board; pos.n = info.pos.n; pos.komi = info.pos.komi
  pos.caps = (Int(info.pos.caps_black), Int(info.pos.caps_white))
  pos.ko = info.pos.ko < 0 ? nothing : from_flat(info.pos.ko + 1, env)
  pos.to_play = info.pos.to_play
  pos.done = info.done != 0
  # `recent`: the C ABI exposes the last two moves of a node (agz_position_info.last_move / prev_move: what select_leaf's
  # pass rule and the double-pass end need, mcts.jl:119-126); the full move list of a game is extract_data's
  # (agz_records_game).  `board_deltas` is not rebuilt here: features come from agz_features / the leaf feature call.
  to_pm(a, color) = PlayerMove(color, a == env.N^2 ? nothing : from_flat(a + 1, env))
  if x.id == root(p)
    pos.recent = copy(getfield(p, :recent))       # the player's root: the game's whole move list (extract_data replays it)
  else
    if info.pos.prev_move >= 0 push!(pos.recent, to_pm(Int(info.pos.prev_move), pos.to_play)) end
    if info.pos.last_move >= 0 push!(pos.recent, to_pm(Int(info.pos.last_move), -pos.to_play)) end
  end
  pos
end

get_position(p::MCTSPlayer) = position(p.root)                                  # mcts_play.jl:141-142

function suggest_move(p::MCTSPlayer)                                            # mcts_play.jl:144-151
  current_readouts = N(p)
  while N(p) < current_readouts + p.num_readouts
    tree_search!(p)
  end
  pick_move(p)
end

function should_resign(p::MCTSPlayer)                                           # mcts_play.jl:124
  r = Ref{Int32}(0)
  check(p.engine, ccall((:agz_tree_should_resign, libagz), Int32, (Ptr{Cvoid}, Int32, Ref{Int32}), p.engine.handle, 0, r))
  r[] != 0
end

function is_done(p::MCTSPlayer)                                                 # mcts_play.jl:120
  p.result != 0 && return true
  r = Ref{Int32}(0)
  check(p.engine, ccall((:agz_tree_is_done, libagz), Int32, (Ptr{Cvoid}, Int32, Int32, Ref{Int32}),
                        p.engine.handle, 0, root(p), r))
  r[] != 0
end

function set_result!(p::MCTSPlayer, winner, was_resign)                         # mcts_play.jl:100-108
  p.result = winner
  p.result_string = was_resign ? (winner == BLACK ? "B+R" : "W+R") : result_string(position(p.root))
end

# replay_position(pos, result), board.jl:557-578: the positions before each move of `recent`, from the empty board
function replay_positions(env::GoEnv, komi, recent::Vector{PlayerMove})
  positions = Position[]
  pos = Position(env; komi = komi)
  for pm in recent
    push!(positions, pos)
    pos = play_move!(pos, pm.move)
  end
  positions, pos
end

# extract_data(player) -> (positions, pis, results), mcts_play.jl:126-139 -- ONE argument, as train.jl:58 calls it
function extract_data(p::MCTSPlayer)
  rootpos = position(p.root)
  length(p.searches_π) == rootpos.n || throw(AssertionError("length(searches_π) == root.position.n"))      # :127
  rootpos.n == length(rootpos.recent) || throw(AssertionError("GoPosition history is incomplete"))          # board.jl:568
  positions, _ = replay_positions(p.env, rootpos.komi, rootpos.recent)
  positions, deepcopy(p.searches_π), fill(p.result, length(positions))
end

# ------------------------------------------------------------------ self-play
# selfplay(env, nn, num_ro) -> the finished game's player (src/selfplay.jl:1-45,:44), exactly the call train() makes
# (train.jl:57).  With `games = G` (ours) the same loop runs for G concurrent games on the device and a Vector of the
# same objects comes back, ordered by game id.
#
# SelfPlayPlayer is the MCTSPlayer of ONE finished game, read-only: the fields train() and extract_data read
# (mcts_play.jl:3-15) -- result, result_string, qs, searches_π, root.position (the final GoPosition with its whole
# `recent`: .n, .board, .caps ...) -- as the device recorded them.  The tree stayed on the device and its slot was
# recycled.  It also carries the record's own fields (game_id, moves as 1-based flat moves, short_searches).
struct FinishedRoot
  position::Position
end
struct SelfPlayPlayer
  env::GoEnv
  network
  num_readouts::Int
  two_player_mode::Bool
  τ_threshold::Int
  qs::Vector{Float32}
  searches_π::Vector{Vector{Float32}}
  result::Int
  result_string::String
  root::FinishedRoot
  resign_threshold::Float64
  position::Position
  positions::Vector{Position}           # the position before each move (replay_position, board.jl:557-578)
  game_id::UInt64
  moves::Vector{Int}                    # 1-based flat moves, N^2+1 = pass
  short_searches::Int                   # moves played on fewer than num_ro readouts (full node pool, agz_config.pool_policy); 0 = the reference's game
end
is_done(p::SelfPlayPlayer) = true
get_position(p::SelfPlayPlayer) = p.root.position

# extract_data(player), mcts_play.jl:126-139: every result entry is the final game result
function extract_data(p::SelfPlayPlayer)
  length(p.searches_π) == p.root.position.n || throw(AssertionError("length(searches_π) == root.position.n"))
  copy(p.positions), deepcopy(p.searches_π), fill(p.result, length(p.positions))
end

# The reference draws from Julia's global RNG (selfplay.jl:9, mcts.jl:133,235, mcts_play.jl:61,66), so successive
# selfplay calls see successive random numbers.  The engine's draws are a function of (seed, game id, move, site)
# (include/agz_draws.h): seed!(s) is Random.seed!(s) for it, and every selfplay call plays the next unused game ids.
const STREAM = Ref((UInt64(0), UInt64(0)))                     # (seed, next game id)
seed!(s::Integer) = (STREAM[] = (UInt64(s), UInt64(0)); nothing)

function selfplay(env::GoEnv, nn::NeuralNet, num_ro::Int = 800; games::Union{Nothing, Int} = nothing,
                  slots::Union{Nothing, Int} = nothing, seed = nothing, game_id_base = nothing)
  G = games === nothing ? 1 : games
  if seed === nothing
    seed, next = STREAM[]
    if game_id_base === nothing
      game_id_base = next
      STREAM[] = (seed, next + UInt64(G))
    end
  end
  game_id_base === nothing && (game_id_base = 0)
  e = Engine(board_size = env.N, tower_height = nn.tower_height, games = slots === nothing ? min(G, 1024) : slots,
             num_readouts = num_ro, seed = seed, game_id_base = game_id_base, record_capacity_games = G + 8)
  copy_weights!(e, nn.engine)
  check(e, ccall((:agz_selfplay_start, libagz), Int32, (Ptr{Cvoid}, Int64), e.handle, G))
  while ccall((:agz_records_count, libagz), Int64, (Ptr{Cvoid},), e.handle) < G
    check(e, ccall((:agz_selfplay_step, libagz), Int32, (Ptr{Cvoid}, Int32), e.handle, 16))
    check_pool(e)     # a game whose node pool ran out cannot finish: raise instead of stepping for ever
  end
  players = SelfPlayPlayer[]
  A = env.action_space
  τ = (env.N * env.N ÷ 12) ÷ 2 * 2
  for k in 0:G-1
    h = Ref{AgzGameHeader}()
    check(e, ccall((:agz_records_header, libagz), Int32, (Ptr{Cvoid}, Int64, Ref{AgzGameHeader}), e.handle, k, h))
    n = Int(h[].num_moves)
    moves = zeros(Int16, max(n, 1)); pis = zeros(Float32, A, max(n, 1)); qs = zeros(Float32, max(n, 1))
    check(e, ccall((:agz_records_game, libagz), Int32, (Ptr{Cvoid}, Int64, Ptr{Int16}, Ptr{Float32}, Ptr{Float32}),
                   e.handle, k, moves, pis, qs))
    rs = h[].was_resign != 0 ? (h[].result == BLACK ? "B+R" : "W+R") : result_string(h[].final_score)
    fmoves = Int.(moves[1:n]) .+ 1
    recent, color = PlayerMove[], BLACK
    for f in fmoves
      push!(recent, PlayerMove(color, from_flat(f, env))); color = -color
    end
    positions, final = replay_positions(env, 7.5, recent)
    push!(players, SelfPlayPlayer(env, nn, num_ro, false, τ, qs[1:n], [pis[:, i] for i in 1:n], Int(h[].result), rs,
                                  FinishedRoot(final), h[].resign_disabled != 0 ? -1.0 : -0.9, final, positions,
                                  h[].game_id, fmoves, Int(h[].short_searches)))
  end
  sort!(players, by = r -> r.game_id)
  games === nothing ? players[1] : players
end

# get_replay_batch(pos_buffer, π_buffer, res_buffer; batch_size), src/train.jl:4-12: batch_size distinct entries,
# π as an A x B matrix
function get_replay_batch(pos_buffer::Vector{Position}, π_buffer, res_buffer; batch_size = 32)
  picks = randperm(length(pos_buffer))[1:batch_size]             # sample(1:n, batch_size, replace = false)
  pos_buffer[picks], reduce(hcat, π_buffer[picks]), res_buffer[picks]
end

# _train(nn, (positions, π, z), opt; epochs), src/neural_net.jl:85-101 (call: train.jl:70) as intended -- the
# reference's does not run at HEAD (SURVEY.md D3): minibatches of 32 positions, each ONE agz_train_step on the device
# (training-mode forward, 0.01 crossentropy + 0.01 mse + 1e-4 sum(θ²), backward, Momentum update of nn in place).
# Returns the summed minibatch loss / epochs (:98-100).  Momentum(η, ρ = 0.9) stands for Flux.Momentum (train.jl:54):
# its velocity lives in the network's engine.
struct Momentum
  eta::Float32
  rho::Float32
end
Momentum(eta = 0.01f0) = Momentum(eta, 0.9f0)

function _train(nn::NeuralNet, input_data, opt::Momentum; epochs = 1)
  positions, π, z = input_data
  env = nn.env; N = env.N; n = length(positions)
  boards = cat(dims = 3, (p.board for p in positions)...)
  deltas = zeros(Int8, N, N, 7, n); nd = zeros(Int32, n)
  for (b, p) in enumerate(positions)
    k = size(p.board_deltas, 3); nd[b] = k
    deltas[:, :, 1:k, b] .= p.board_deltas
  end
  tp = Int8[p.to_play for p in positions]
  feats = zeros(Float32, N * N * 17, n)
  check(nn.engine, ccall((:agz_features, libagz), Int32,
        (Ptr{Cvoid}, Ptr{Int8}, Ptr{Int8}, Ptr{Int32}, Ptr{Int8}, Int32, Ptr{Float32}),
        nn.engine.handle, boards, deltas, nd, tp, n, feats))
  cuts = vcat(collect(1:32:n), n + 1)
  length(cuts) > 2 && cuts[end] - cuts[end-1] == 1 && deleteat!(cuts, length(cuts) - 1)   # BatchNorm needs two rows
  loss_avg = 0f0
  for _ in 1:epochs, j in 1:length(cuts)-1
    r = cuts[j]:cuts[j+1]-1
    loss_avg += train_step!(nn.engine, feats[:, r], Matrix{Float32}(π[:, r]), Float32.(z[r]); eta = opt.eta, rho = opt.rho)[1]
  end
  loss_avg / epochs
end

# every (layer, kind) of a network with `t` residual blocks (ids as in include/agz.h)
function layer_kinds(t::Integer)
  lk = Tuple{Int32, Int32}[]
  for l in vcat(collect(0:2t), [-1, -2]), k in 0:6
    push!(lk, (l, k))
  end
  for l in (-3, -4, -5), k in 0:1
    push!(lk, (l, k))
  end
  lk
end

function copy_weights!(dst::Engine, src::Engine)
  for (l, k) in layer_kinds(src.cfg.tower_height)
    n = ccall((:agz_net_param_count, libagz), Int64, (Ptr{Cvoid}, Int32, Int32), src.handle, l, k)
    buf = zeros(Float32, n)
    check(src, ccall((:agz_net_get_weights, libagz), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{Float32}, Int64),
                     src.handle, l, k, buf, n))
    check(dst, ccall((:agz_net_set_weights, libagz), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{Float32}, Int64),
                     dst.handle, l, k, buf, n))
  end
  dst
end

# tower arithmetic of an engine's network: :f32 (default, exact) or :f16 (fp16 operands, f32 accumulate)
# the f32 Winograd tower as one persistent launch (same bits as one launch per layer, include/agz.h)
set_tower_persistent!(e::Engine, on::Bool) =
  check(e, ccall((:agz_net_set_tower_persistent, libagz), Int32, (Ptr{Cvoid}, Int32), e.handle, on ? 1 : 0))
# the F(4x4,3x3) tower (boards >= 13x13) as two independent layer chains on two streams (default) or one
set_tower_streams!(e::Engine, n::Integer) =
  check(e, ccall((:agz_net_set_tower_streams, libagz), Int32, (Ptr{Cvoid}, Int32), e.handle, n))
set_precision!(e::Engine, p::Symbol) =      # :f32 (default, exact), :f16 (fp16 operands), :f32s (f32 as split f16 operands)
  check(e, ccall((:agz_net_set_precision, libagz), Int32, (Ptr{Cvoid}, Int32), e.handle, p === :f16 ? 1 : p === :f32s ? 2 : 0))

# ------------------------------------------------------------------ evaluate
# evaluate(env, black_net, white_net; num_games, ro), src/neural_net.jl:103-158: both networks live
# in one arena_mode engine (network 0 = Black's, 1 = White's), every game is a pair of slots, all
# games run concurrently.  Black's tally is `result(black.root.position) == BLACK` (:147), i.e.
# final_score > 0, also for resigned games.
function evaluate(env::GoEnv, black_net::NeuralNet, white_net::NeuralNet; num_games = 400, ro = 800,
                  verbose::Bool = false, seed = 0, pairs::Int = min(num_games, 512))
  @assert black_net.tower_height == white_net.tower_height
  e = Engine(board_size = env.N, tower_height = black_net.tower_height, games = 2 * pairs, num_readouts = ro,
             seed = seed, record_capacity_games = num_games + 8, arena_mode = true)
  copy_weights!(e, black_net.engine)
  check(e, ccall((:agz_net_select, libagz), Int32, (Ptr{Cvoid}, Int32), e.handle, 1))
  copy_weights!(e, white_net.engine)
  check(e, ccall((:agz_net_select, libagz), Int32, (Ptr{Cvoid}, Int32), e.handle, 0))
  check(e, ccall((:agz_selfplay_start, libagz), Int32, (Ptr{Cvoid}, Int64), e.handle, num_games))
  while ccall((:agz_records_count, libagz), Int64, (Ptr{Cvoid},), e.handle) < num_games
    check(e, ccall((:agz_selfplay_step, libagz), Int32, (Ptr{Cvoid}, Int32), e.handle, 16))
    check_pool(e)
  end
  games_won = 0
  for k in 0:num_games-1
    h = Ref{AgzGameHeader}()
    check(e, ccall((:agz_records_header, libagz), Int32, (Ptr{Cvoid}, Int64, Ref{AgzGameHeader}), e.handle, k, h))
    games_won += h[].final_score > 0
  end
  verbose && print("Won $games_won / $num_games. Win rate: $(games_won/num_games). ")
  return games_won / num_games ≥ 0.55
end

# ------------------------------------------------------------------ replay batches
# get_replay_batch(pos_buffer, π_buffer, res_buffer; batch_size), src/train.jl:4-12, with the
# positions kept as move lists: `games[g]` is a SelfPlayPlayer, a sample is (g, ply) with ply = 0 the
# empty board.  Returns the N x N x 17 x B feature tensor get_feats would build, π (A x B), results.
function get_replay_batch(e::Engine, env::GoEnv, games::Vector{SelfPlayPlayer}, samples::Vector{Tuple{Int,Int}})
  used = sort(unique(first.(samples)))
  offs = Dict{Int,Int32}(); moves = Int16[]
  for g in used
    offs[g] = length(moves)
    append!(moves, Int16.(games[g].moves .- 1))
  end
  B = length(samples)
  off = Int32[offs[g] for (g, _) in samples]; ply = Int32[j for (_, j) in samples]
  feats = zeros(Float32, env.N, env.N, 17, B)
  check(e, ccall((:agz_replay_features, libagz), Int32,
                 (Ptr{Cvoid}, Ptr{Int16}, Int64, Ptr{Int32}, Ptr{Int32}, Int32, Ptr{Float32}, Int32),
                 e.handle, moves, length(moves), off, ply, B, feats, 0))
  π = hcat((games[g].searches_π[j + 1] for (g, j) in samples)...)
  feats, π, [games[g].result for (g, _) in samples]
end

# ---- replay arena, RCCL exchange and the training step (train.jl:47-74; SURVEY.md 8e, 8f rows 1 and 4)
# Every rank's finished games are all-gathered by libagz itself (RCCL over xGMI, device to device) into a
# device-resident replay arena; (game, ply) samples become (features, pi, z) on the device; one call is one
# optimisation step of _train.  The 128-byte communicator id is generated on rank 0 and handed to the other
# ranks by the host (Distributed.jl, MPI.jl, a shared file -- the library does not care).

comm_unique_id() = (id = zeros(UInt8, 128);
                    st = ccall((:agz_comm_unique_id, libagz), Int32, (Ptr{UInt8},), id);
                    st == AGZ_OK || error("libagz status $st: " * unsafe_string(ccall((:agz_last_error, libagz), Cstring, (Ptr{Cvoid},), C_NULL)));
                    id)

function comm_create(e::Engine, rank::Integer, world::Integer, id::Vector{UInt8})
  c = Ref{Ptr{Cvoid}}(C_NULL)
  check(e, ccall((:agz_comm_create, libagz), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{UInt8}, Ref{Ptr{Cvoid}}),
                 e.handle, rank, world, id, c))
  c[]
end
comm_destroy(c::Ptr{Cvoid}) = ccall((:agz_comm_destroy, libagz), Cvoid, (Ptr{Cvoid},), c)

# push_data.(buffers, extract_data(player)) for the games of EVERY rank (train.jl:57-62); comm = C_NULL on one GPU
function allgather_records!(e::Engine, comm::Ptr{Cvoid} = C_NULL)
  added = Ref{Int64}(0)
  check(e, ccall((:agz_allgather_records, libagz), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Int64}), e.handle, comm, added))
  check(e, ccall((:agz_records_clear, libagz), Int32, (Ptr{Cvoid},), e.handle))
  added[]
end
# The same exchange with the bytes carried by the HOST's communication library (MPI.jl, Distributed): `allgather` is any
# function that takes this rank's Vector and returns the concatenation of every rank's, rank order.  Everything but the
# two collectives is libagz: what this rank announces, the count check / chunk stride (agz_gather_plan: the function
# agz_allgather_records itself calls), the packed export, device-side indexing and compaction into the arena.
# A rank whose pack step fails still joins the count collective with {-1, status}: every rank errors together.
function allgather_records_hosted!(e::Engine, world::Integer, allgather::Function)
  nb = Ref{Int64}(0)
  st = ccall((:agz_records_packed_size, libagz), Int32, (Ptr{Cvoid}, Ref{Int64}), e.handle, nb)
  mine = st == AGZ_OK ? Int64[ccall((:agz_records_count, libagz), Int64, (Ptr{Cvoid},), e.handle), nb[]] : Int64[-1, st]
  counts = allgather(mine)::Vector{Int64}                                        # 2 x Int64 per rank
  stride = Ref{Int64}(0); total = Ref{Int64}(0)
  pst = ccall((:agz_gather_plan, libagz), Int32, (Ptr{Int64}, Int32, Ref{Int64}, Ref{Int64}), counts, world, stride, total)
  st == AGZ_OK || check(e, st)
  pst == AGZ_OK || error("libagz status $pst: " * unsafe_string(ccall((:agz_last_error, libagz), Cstring, (Ptr{Cvoid},), C_NULL)))
  total[] == 0 && return 0
  send = zeros(UInt8, stride[])
  check(e, ccall((:agz_records_export_packed, libagz), Int32, (Ptr{Cvoid}, Ptr{UInt8}, Int64, Int32), e.handle, send, mine[2], 0))
  recv = allgather(send)::Vector{UInt8}                                          # world x stride bytes
  added = Ref{Int64}(0)
  check(e, ccall((:agz_replay_ingest_gathered, libagz), Int32,
                 (Ptr{Cvoid}, Ptr{UInt8}, Int32, Int32, Int64, Ptr{Int64}, Ref{Int64}),
                 e.handle, recv, 0, world, stride[], counts, added))
  check(e, ccall((:agz_records_clear, libagz), Int32, (Ptr{Cvoid},), e.handle))
  added[]
end
broadcast_weights!(e::Engine, comm::Ptr{Cvoid}, root::Integer = 0) =
  check(e, ccall((:agz_broadcast_weights, libagz), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{Int64}), e.handle, comm, root, C_NULL))

replay_games(e::Engine) = ccall((:agz_replay_count, libagz), Int64, (Ptr{Cvoid},), e.handle)
replay_positions(e::Engine) = ccall((:agz_replay_positions, libagz), Int64, (Ptr{Cvoid},), e.handle)     # length(pos_buffer)
replay_trim!(e::Engine, memory_size::Integer) =                                                         # shrink, train.jl:52
  check(e, ccall((:agz_replay_trim, libagz), Int32, (Ptr{Cvoid}, Int64), e.handle, memory_size))
function replay_game_lengths(e::Engine)
  h = Ref{AgzGameHeader}()
  [begin
     check(e, ccall((:agz_replay_header, libagz), Int32, (Ptr{Cvoid}, Int64, Ref{AgzGameHeader}), e.handle, k, h))
     Int(h[].num_moves)
   end for k in 0:replay_games(e)-1]
end

# get_replay_batch (train.jl:4-12): `games`/`plies` are 0-based (game index in the arena, move number);
# returns (feats N*N*17 x B, pi A x B, z B) -- column-major, i.e. exactly what `_train` consumes
function replay_batch(e::Engine, env::GoEnv, games::Vector{Int64}, plies::Vector{Int32})
  B = length(games)
  feats = zeros(Float32, env.N * env.N * 17, B); pi = zeros(Float32, env.action_space, B); z = zeros(Float32, B)
  check(e, ccall((:agz_replay_batch, libagz), Int32,
                 (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int32}, Int32, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Int32),
                 e.handle, games, plies, B, feats, pi, z, 0))
  feats, pi, z
end

# _train(nn, (positions, pi, z), Momentum(2f-2)) for one batch (neural_net.jl:85-101): returns
# (total, policy, value, regulariser) losses before the update
function train_step!(e::Engine, feats::Matrix{Float32}, pi::Matrix{Float32}, z::Vector{Float32}; eta = 0.02f0, rho = 0.9f0)
  losses = zeros(Float32, 4)
  check(e, ccall((:agz_train_step, libagz), Int32,
                 (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Int32, Int32, Float32, Float32, Ptr{Float32}),
                 e.handle, feats, pi, z, length(z), 0, eta, rho, losses))
  Tuple(losses)
end

end # module
