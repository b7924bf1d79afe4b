// Enumerations of include/rodeo_kalman.h needed by device code (repeated here so that hiprtc translation units do not
// need the C header; the values are checked against the header in api.hip).
#pragma once
#ifndef RK_INTERROGATE_RODEO
#define RK_INTERROGATE_RODEO      0
#define RK_INTERROGATE_SCHOBER    1
#define RK_INTERROGATE_KRAMER     2
#define RK_INTERROGATE_CHKREBTII  3
#endif

// Optional members of a right-hand side type (csrc/rhs.hpp): user-supplied types need not declare them.
namespace rk {
template <class...> using void_t_ = void;
template <class R, class = void> struct rhs_has_tile_form { static constexpr bool value = false; };
template <class R> struct rhs_has_tile_form<R, void_t_<decltype(R::HAS_TILE_FORM)>> { static constexpr bool value = R::HAS_TILE_FORM; };
// number of tile-form coefficients (an array of that size may be declared for any right-hand side)
template <class R, class = void> struct tile_form_consts { static constexpr int N = 1; };
template <class R> struct tile_form_consts<R, void_t_<decltype(R::NTILEK)>> { static constexpr int N = R::NTILEK; };
template <class R, class = void> struct rhs_has_tile3_form { static constexpr bool value = false; };
template <class R> struct rhs_has_tile3_form<R, void_t_<decltype(R::HAS_TILE3_FORM)>> { static constexpr bool value = R::HAS_TILE3_FORM; };
template <class R, class = void> struct rhs_has_fjac0 { static constexpr bool value = false; };
template <class R> struct rhs_has_fjac0<R, void_t_<decltype(R::HAS_FJAC0)>> { static constexpr bool value = R::HAS_FJAC0; };
}  // namespace rk
