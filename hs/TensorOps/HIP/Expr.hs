-- | Reifying the closures of 'liftT' / 'liftB'.
--
-- @liftT :: (Vec n (ElemT t) -> ElemT t) -> Vec n (t o) -> t o@ (@src/TensorOps/Types.hs:56-59@) takes an opaque
-- Haskell function, which a GPU cannot run.  But the backend chooses @ElemT@, and at the @TOp@ level every such
-- function is @forall a. RealFloat a => ...@ (@VFunc@, @Types.hs:114-117@; @TO.map f = map' f (diff f)@ instantiates
-- @f@ at @ad@'s @Forward a@, @src/TensorOps/TOp.hs:209-213@).  So the element type of this backend is 'E': either a
-- concrete number or a symbolic expression.  Applying the closure to symbolic variables yields its expression
-- tree, which is flattened (with common-subexpression elimination) into the SSA program @to_expr_compile@ takes
-- and cached by structure.  The library then classifies the program against its pre-fused kernels by evaluating
-- it on random points, so the order in which @ad@ happened to build the derivative does not matter.
--
-- NOT type-checked in this repository's build image (no GHC).  The same construction runs, tested, in
-- @tensor-ops_amd/host/tensorops/expr.hpp@ (C++) and @tensor-ops_amd/hipt.py@ (Python).
module TensorOps.HIP.Expr
  ( E(..), X(..), reify, compileX, evalX
  ) where

import           Control.Monad.State.Strict
import           Data.IORef
import           Data.Int
import           Foreign
import           Foreign.C.Types
import           System.IO.Unsafe           (unsafePerformIO)
import           TensorOps.HIP.FFI
import qualified Data.Map.Strict            as M

-- | Opcodes of @include/tensorops_hip.h@ (@TO_X_*@), in the header's order.
data Op = OConst | OAdd | OSub | OMul | ODiv | ONeg | ORecip | OExp | OLog | OSqrt | OAbs | OSignum
        | OSin | OCos | OTanh | OPow | OMax | OMin
  deriving (Eq, Ord, Enum, Show)

-- | Expression trees over the closure's arguments.
data X = XVar !Int | XConst !Double | X1 !Op X | X2 !Op X X
  deriving (Eq, Ord, Show)

-- | @ElemT HipT@ / @ElemB HipB@.  Scalars that cross the class boundary as numbers (@scaleT@'s factor, the
-- results of @dot@ / @(!)@) are 'C'; inside a closure being reified they are 'S'.
data E = C !Double | S X

lift1 :: Op -> (Double -> Double) -> E -> E
lift1 _ f (C a) = C (f a)
lift1 o _ (S a) = S (X1 o a)

lift2 :: Op -> (Double -> Double -> Double) -> E -> E -> E
lift2 _ f (C a) (C b) = C (f a b)
lift2 o _ a     b     = S (X2 o (toX a) (toX b))

toX :: E -> X
toX (C a) = XConst a
toX (S x) = x

concrete :: String -> E -> Double
concrete _   (C a) = a
concrete who (S _) = error ("tensorops_hip: " ++ who ++ " of a symbolic element: the closure inspects its argument "
                           ++ "(comparison / rounding), which cannot be reified; use max / min / abs / signum")

instance Show E where
  show (C a) = show a
  show (S x) = show x

instance Num E where
  (+) = lift2 OAdd (+)
  (-) = lift2 OSub (-)
  (*) = lift2 OMul (*)
  negate = lift1 ONeg negate
  abs    = lift1 OAbs abs
  signum = lift1 OSignum signum
  fromInteger = C . fromInteger

instance Fractional E where
  (/)   = lift2 ODiv (/)
  recip = lift1 ORecip recip
  fromRational = C . fromRational

instance Floating E where
  pi    = C pi
  exp   = lift1 OExp exp
  log   = lift1 OLog log
  sqrt  = lift1 OSqrt sqrt
  sin   = lift1 OSin sin
  cos   = lift1 OCos cos
  tanh  = lift1 OTanh tanh
  (**)  = lift2 OPow (**)
  tan x   = sin x / cos x
  sinh x  = (exp x - exp (negate x)) / 2
  cosh x  = (exp x + exp (negate x)) / 2
  asin  = concreteOnly "asin" asin
  acos  = concreteOnly "acos" acos
  atan  = concreteOnly "atan" atan
  asinh x = log (x + sqrt (x * x + 1))
  acosh x = log (x + sqrt (x * x - 1))
  atanh x = log ((1 + x) / (1 - x)) / 2

concreteOnly :: String -> (Double -> Double) -> E -> E
concreteOnly who f = C . f . concrete who

-- | Equality and order exist on concrete values (@argMax@, the classifiers of the apps); on symbolic values only
-- the operations with a kernel-side meaning are defined.
instance Eq E where
  a == b = concrete "(==)" a == concrete "(==)" b
instance Ord E where
  compare a b = compare (concrete "compare" a) (concrete "compare" b)
  max = lift2 OMax max
  min = lift2 OMin min
instance Real E where
  toRational = toRational . concrete "toRational"
instance RealFrac E where
  properFraction e = let (n, f) = properFraction (concrete "properFraction" e) in (n, C f)
instance RealFloat E where
  floatRadix     = floatRadix     . concrete "floatRadix"
  floatDigits    = floatDigits    . concrete "floatDigits"
  floatRange     = floatRange     . concrete "floatRange"
  decodeFloat    = decodeFloat    . concrete "decodeFloat"
  encodeFloat m  = C . encodeFloat m
  isNaN          = isNaN          . concrete "isNaN"
  isInfinite     = isInfinite     . concrete "isInfinite"
  isDenormalized = isDenormalized . concrete "isDenormalized"
  isNegativeZero = isNegativeZero . concrete "isNegativeZero"
  isIEEE _       = True

-- | Evaluate a tree on concrete arguments (the @n = 0@ case of 'liftT', and a reference for tests).
evalX :: [Double] -> X -> Double
evalX args = go
  where
    go (XVar i)    = args !! i
    go (XConst c)  = c
    go (X1 o a)    = un o (go a)
    go (X2 o a b)  = bin o (go a) (go b)
    un ONeg = negate; un ORecip = recip; un OExp = exp; un OLog = log; un OSqrt = sqrt; un OAbs = abs
    un OSignum = signum; un OSin = sin; un OCos = cos; un OTanh = tanh
    un o = error ("evalX: not unary: " ++ show o)
    bin OAdd = (+); bin OSub = (-); bin OMul = (*); bin ODiv = (/); bin OPow = (**); bin OMax = max; bin OMin = min
    bin o = error ("evalX: not binary: " ++ show o)

-- | The SSA program of a tree: value @v < arity@ is input @v@, value @arity + i@ the result of instruction @i@
-- = (op, a, b); constants are pooled.  Structurally equal subtrees become one value.
data Prog = Prog { pCode :: [(Int32, Int32, Int32)], pConsts :: [Double] }

flatten :: Int -> X -> Prog
flatten arity x0 =
    let (r, (_, code, consts, _)) = runState (go x0) (M.empty, [], [], M.empty)
        n      = length code
        code'  = reverse code
        -- the result must be the LAST value: a bare variable or an earlier value gets `+ 0`
        final | n > 0 && r == arity + n - 1 = Prog code' (reverse consts)
              | otherwise =
                  let ci = length consts
                      zc = (fromIntegral (fromEnum OConst), fromIntegral ci, 0)
                      ad = (fromIntegral (fromEnum OAdd), fromIntegral r, fromIntegral (arity + n))
                  in Prog (code' ++ [zc, ad]) (reverse consts ++ [0])
    in final
  where
    emit key ins = do
      (seen, code, consts, cm) <- get
      let v = arity + length code
      put (M.insert key v seen, ins : code, consts, cm)
      return v
    go :: X -> State (M.Map X Int, [(Int32, Int32, Int32)], [Double], M.Map Double Int) Int
    go (XVar i) = return i
    go x = do
      (seen, _, _, _) <- get
      case M.lookup x seen of
        Just v  -> return v
        Nothing -> case x of
          XConst c -> do
            (s, code, consts, cm) <- get
            ci <- case M.lookup c cm of
                    Just i  -> return i
                    Nothing -> do
                      let i = length consts
                      put (s, code, c : consts, M.insert c i cm)
                      return i
            emit x (fromIntegral (fromEnum OConst), fromIntegral ci, 0)
          X1 o a -> do
            va <- go a
            emit x (fromIntegral (fromEnum o), fromIntegral va, fromIntegral va)
          X2 o a b -> do
            va <- go a
            vb <- go b
            emit x (fromIntegral (fromEnum o), fromIntegral va, fromIntegral vb)
          XVar _ -> error "unreachable"

{-# NOINLINE exprCache #-}
exprCache :: IORef (M.Map (Int, X) HX)
exprCache = unsafePerformIO (newIORef M.empty)

-- | Compile (once per structure) the program of an @arity@-ary tree.
compileX :: Int -> X -> HX
compileX arity x = unsafePerformIO $ do
    m <- readIORef exprCache
    case M.lookup (arity, x) m of
      Just h  -> return h
      Nothing -> do
        let Prog code consts = flatten arity x
            flat = concat [ [o, a, b] | (o, a, b) <- code ]
        h <- hipReady `seq`
             withArrayLen flat $ \_ pc ->
             withArrayLen (map realToFrac consts) $ \nc pk ->
             alloca $ \out -> do
               chk (c_expr_compile (fromIntegral arity) (fromIntegral (length code)) pc
                                   (fromIntegral nc) (pk :: Ptr CDouble) out)
               peek out >>= newForeignPtr p_expr_release
        atomicModifyIORef' exprCache (\mm -> (M.insert (arity, x) h mm, ()))
        return h

-- | Apply a closure to @n@ symbolic variables.  'Left' a constant when the result does not depend on them.
reify :: Int -> ([E] -> E) -> Either Double (X, HX)
reify n f = case f [ S (XVar i) | i <- [0 .. n - 1] ] of
    C c -> Left c
    S x -> Right (x, compileX n x)
